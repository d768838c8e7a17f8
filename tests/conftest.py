import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
# one hardware queue per stream of the handles in flight (a deployment knob of the HIP runtime; bench.py and the
# tools set it the same way): must be in the environment before HIP starts
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def opj():
    """A real libopenjp2 driven like the reference drives it; skip where none is installed."""
    from oracle.oracle import OpjReplay
    try:
        return OpjReplay()
    except OSError as e:
        pytest.skip(f"libopenjp2 replay unavailable: {e}")


def golden_case(golden, name):
    from j2k_amd import synth
    from oracle.oracle import make_params
    g = golden[name]
    pl = synth.planes(g["width"], g["height"], g["ncomp"], g["prec"], g["seed"], g["dist"])
    p = make_params(g["width"], g["height"], g["ncomp"], g["prec"], **g["params"])
    path = os.path.join(GOLDEN_DIR, name + ".j2k")
    cs = open(path, "rb").read() if os.path.exists(path) else None
    return g, pl, p, cs

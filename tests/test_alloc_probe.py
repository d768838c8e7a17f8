"""Host-side layer allocation (rate control) on REAL Tier-1 results, no GPU: tests/golden/
alloc_2048_rgb16_97.bin.gz holds what the Tier-1 kernels produced for a 2048 x 2048 RGB16 9/7 frame (per-pass byte
counts and distortion sums of its 3072 code-blocks; written by a -DJ2K_ALLOC_DUMP build, tools/README.md).
tools/alloc_probe.cpp runs rate_control.cpp's allocation -- settled blocks, slope bounds, candidates that certainly fit,
the layer-by-layer pricer -- and OpenJPEG's plain procedure (every round scans and prices everything) on it and exits
non-zero unless the two allocations are identical."""
import gzip
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RATIOS = [["20"], ["100"], ["1000"], ["200", "50", "20", "10", "5"], ["8", "4", "2", "1.3"], ["30", "29", "28"], ["2"], ["40", "10", "0"],
          ["psnr", "30"], ["psnr", "25", "35", "45"], ["psnr", "60"], ["psnr", "38", "0"]]


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    d = tmp_path_factory.mktemp("alloc_probe")
    csrc = os.path.join(ROOT, "j2k_amd", "csrc")
    exe = str(d / "alloc_probe")
    srcs = [os.path.join(ROOT, "tools", "alloc_probe.cpp")] + [os.path.join(csrc, f) for f in ("geometry.cpp", "tier2.cpp", "jp2.cpp", "rate_control.cpp", "workers.cpp")]
    build = subprocess.run(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), *srcs, "-lpthread", "-o", exe], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    dump = str(d / "alloc.bin")
    with gzip.open(os.path.join(ROOT, "tests", "golden", "alloc_2048_rgb16_97.bin.gz"), "rb") as f, open(dump, "wb") as o:
        shutil.copyfileobj(f, o)
    return exe, dump


@pytest.mark.parametrize("ratios", RATIOS, ids=lambda r: "_".join(r))
def test_fast_allocation_equals_plain_procedure_on_real_tier1_results(probe, ratios):
    exe, dump = probe
    run = subprocess.run([exe, dump, *ratios], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.stdout[-1500:], run.stderr[-1500:])
    assert "same allocation" in run.stdout


@pytest.mark.parametrize("min_scan", ["0", "300"])
@pytest.mark.parametrize("ratios", [r for r in RATIOS if r[0] != "psnr"], ids=lambda r: "_".join(r))
def test_allocation_through_the_device_interface_equals_plain_procedure(probe, ratios, min_scan):
    """The per-block work (bounds, the walk over the thresholds ahead, the scans of rounds with at least `min_scan` open
    blocks) goes through RateDevice -- here its host stand-in, block by block through rate_block.h as rate.hip's kernels go."""
    exe, dump = probe
    run = subprocess.run([exe, dump, *ratios], capture_output=True, text=True, timeout=300, env=dict(os.environ, J2K_PROBE_DEVICE=min_scan))
    assert run.returncode == 0, (run.stdout[-1500:], run.stderr[-1500:])
    assert "same allocation" in run.stdout

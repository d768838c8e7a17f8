"""CPU: the lane-per-block Tier-1 decoder's per-lane state machine (j2k_amd/csrc/t1_dec_lane.h -- the source the HIP
kernel t1_decode_lanes_kernel runs in each of its 64 lanes) built for the host with one lane and held to the oracle's
block decoder (oracle/j2k_oracle_dec.c: j2ko_t1_decode_block, itself pinned to libopenjp2's decoded samples)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from j2k_amd import synth


def _build(tmp_path_factory, name, extra=()):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    so = str(tmp_path_factory.mktemp(name) / "libt1lane_host.so")
    src = os.path.join(ROOT, "tests", "native", "t1_lane_host.cpp")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-Wall", "-Wno-unknown-pragmas", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=all", *extra, src, "-o", so], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # (an ASan-instrumented library in a plain python: run the decodes in a child that preloads the runtime)
    return so


@pytest.fixture(scope="module")
def lane(tmp_path_factory):
    return _build(tmp_path_factory, "t1lane")


@pytest.fixture(scope="module")
def lane_no_refill(tmp_path_factory):
    """The same decoder with the ring's refill switched off: every byte past the first 64 comes through the slow path that
    only a stream eating two bytes per decision would ever take."""
    return _build(tmp_path_factory, "t1lane_slow", ["-DT1L_TEST_NO_REFILL"])


def _run_cases(so, cases):
    """cases: list of (bytes, w, h, orient, numbps, npasses) -> list of int32 arrays, computed in a child process with libasan preloaded."""
    import pickle
    import sys
    code = r'''
import ctypes as C, pickle, sys
import numpy as np
so, cases = pickle.load(sys.stdin.buffer)
L = C.CDLL(so)
L.t1lane_host_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
outs = []
for data, w, h, o, nb, npass in cases:
    buf = np.frombuffer(data + b"\0", dtype=np.uint8)[:len(data)].copy() if data else np.zeros(1, dtype=np.uint8)
    out = np.zeros((h, w), dtype=np.int32)
    rc = L.t1lane_host_decode(buf.ctypes.data, len(data), w, h, o, nb, npass, out.ctypes.data)
    outs.append((rc, out))
pickle.dump(outs, sys.stdout.buffer)
'''
    asan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", code], input=pickle.dumps((so, cases)), capture_output=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-4000:]
    return pickle.loads(r.stdout)


def _blocks(oracle, rev, prec, seed, dist, sizes):
    pl = synth.planes(200, 136, 1, prec, seed, dist)[0] - (1 << (prec - 1))
    coef = oracle.dwt53(pl, 2) if rev else oracle.dwt97(pl.astype(np.float32), 2)
    rng = np.random.default_rng(seed)
    out = []
    for (bw, bh) in sizes:
        x, y = int(rng.integers(0, 200 - bw)), int(rng.integers(0, 136 - bh))
        blk = coef[y:y + bh, x:x + bw]
        if rev:
            data = (blk.astype(np.int64) << 6).astype(np.int32)
        else:
            data = np.array([[oracle.L.j2ko_quant97(float(v), 0.37) for v in row] for row in blk], dtype=np.int32)
        for orient in (0, 1, 3):
            enc = oracle.t1_block(data, orient)
            if enc["numbps"] and enc["npasses"]:
                out.append((enc["data"], bw, bh, orient, enc["numbps"], enc["npasses"]))
    return out


def test_lane_decoder_equals_the_oracle_block_decoder(lane, oracle):
    sizes = [(64, 64), (64, 64), (37, 64), (64, 13), (5, 7), (1, 64), (64, 1), (4, 4), (33, 31), (64, 62)]
    cases = []
    for rev, prec, dist in ((True, 8, "A"), (False, 16, "A"), (True, 12, "B"), (False, 10, "B")):
        cases += _blocks(oracle, rev, prec, 4000 + prec, dist, sizes)
    # blocks cut short by a rate allocation: every possible last pass of a few blocks, and a truncated codeword segment
    extra = []
    for data, w, h, o, nb, npass in cases[:6]:
        for cut in range(1, npass):
            extra.append((data, w, h, o, nb, cut))
        extra.append((data[:len(data) // 2], w, h, o, nb, npass))
        extra.append((b"", w, h, o, nb, min(npass, 4)))
    cases += extra
    got = _run_cases(lane, cases)
    assert len(got) == len(cases) > 150
    for (data, w, h, o, nb, npass), (rc, out) in zip(cases, got):
        assert rc == 0
        ref = oracle.t1_decode_block(data, w, h, o, nb, npass)
        assert np.array_equal(out, ref.reshape(h, w)), (w, h, o, nb, npass, len(data))


def test_slow_byte_path_decodes_the_same(lane_no_refill, oracle):
    cases = _blocks(oracle, False, 16, 4321, "A", [(64, 64), (33, 31), (5, 7)]) + _blocks(oracle, True, 8, 4322, "A", [(64, 64)])
    got = _run_cases(lane_no_refill, cases)
    assert len(got) == len(cases) >= 10
    for (data, w, h, o, nb, npass), (rc, out) in zip(cases, got):
        assert rc == 0 and len(data) > 64 or w * h < 64 * 64
        assert np.array_equal(out, oracle.t1_decode_block(data, w, h, o, nb, npass).reshape(h, w)), (w, h, o, nb, npass)

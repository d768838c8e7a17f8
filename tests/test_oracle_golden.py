"""CPU: pin the oracle (oracle/j2k_oracle.c) to the committed golden vectors and, where a
libopenjp2 is installed, to the live library driven through the reference's call sequence
(reference: src/common/j2k_openjpeg_codec.cpp:598-750)."""
import hashlib

import numpy as np
import pytest

from conftest import golden_case
from j2k_amd import synth
from oracle.oracle import make_params, strip_com

SMALL = ["g1_64x64_grey_1lvl", "g1_64x64_grey_5lvl", "g2_c1_512_grey_53", "g3_300x200_rgb8_53_rct",
         "g4_300x200_rgb16_53_rct_tile128", "g5_300x200_rgb8_53_ref_literal", "g6_300x200_rgb8_97_ict",
         "g6_300x200_rgb16_97_ict", "g7_300x200_rgb10_53", "g9_300x200_rgba8_53_rct",
         "g9_97x61_grey12_97_4lvl", "g9_150x130_rgb8_97_tile64"]


@pytest.mark.parametrize("name", SMALL)
def test_oracle_matches_golden_codestream(oracle, golden, name):
    g, pl, p, cs = golden_case(golden, name)
    ours = oracle.encode(pl, p)  # no COM
    assert len(ours) == g["length"]
    assert hashlib.sha256(ours).hexdigest() == g["sha256"]
    assert ours == cs


@pytest.mark.parametrize("name", ["g8_c1", "g8_c2", "g8_c3", "g8_c3_5lvl", "g8_c4", "g8_c5"])
def test_oracle_main_headers(oracle, golden, name):
    g = golden[name]
    pl = synth.planes(g["width"], g["height"], g["ncomp"], g["prec"], 1, "B")
    p = make_params(g["width"], g["height"], g["ncomp"], g["prec"], **g["params"])
    ours = oracle.encode(pl, p)
    assert ours[:ours.index(b"\xff\x90")].hex() == g["main_header_hex"]


def test_promote_demote(oracle):
    # reference: src/aftereffects/FrameSeq.cpp:311-314 and :265-268
    for v in (0, 1, 16383, 16384, 16385, 32767, 32768):
        pv = oracle.L.j2ko_promote(v)
        assert pv == (((v - 1) << 1) + 1 if v > 16384 else v << 1)
        assert oracle.L.j2ko_demote(pv) == v
    assert oracle.L.j2ko_promote(32768) == 65535


def test_copy_channel_depth_conversions(oracle):
    # reference: src/common/j2k_codec.cpp:254-371
    rng = np.random.default_rng(1)
    a16 = rng.integers(0, 65536, size=(5, 7, 4), dtype=np.uint16)
    buf = a16.view(np.uint8).reshape(-1)
    for prec in (16, 12, 10):
        got = oracle.copy_channel(buf, 2, 7, 5, 8, 7 * 8, 2, 16, prec)
        assert np.array_equal(got, a16[:, :, 1].astype(np.int32) >> (16 - prec))
    a8 = rng.integers(0, 256, size=(5, 7, 4), dtype=np.uint8)
    buf8 = a8.reshape(-1)
    got = oracle.copy_channel(buf8, 3, 7, 5, 4, 28, 1, 8, 8)
    assert np.array_equal(got, a8[:, :, 3])
    got = oracle.copy_channel(buf8, 1, 7, 5, 4, 28, 1, 8, 12)  # bit replication up-shift
    v = a8[:, :, 1].astype(np.int32)
    assert np.array_equal(got, (v << 4) | (v >> 4))
    got = oracle.copy_channel(buf8, 1, 7, 5, 4, 28, 1, 8, 4)
    assert np.array_equal(got, v >> 4)


def test_dwt53_roundtrip_property(oracle):
    """5/3 forward followed by a textbook inverse restores the input (odd sizes, odd origins)."""
    rng = np.random.default_rng(7)
    for (w, h, x0, y0) in [(37, 21, 0, 0), (64, 64, 0, 0), (33, 18, 5, 3), (1, 9, 1, 1), (8, 1, 3, 0)]:
        a = rng.integers(-2000, 2000, size=(h, w), dtype=np.int32)
        f = oracle.dwt53(a, 1, x0, y0)
        assert np.array_equal(_idwt53_level(f, x0, y0), a)


def _inv53_line(lo, hi, cas):
    n = len(lo) + len(hi)
    x = np.zeros(n, dtype=np.int64)
    if n == 1:
        return np.array([hi[0] // 2] if cas else [lo[0]], dtype=np.int64)
    x[cas::2] = lo
    x[1 - cas::2] = hi
    ext = lambda i: -i if i < 0 else (2 * (n - 1) - i if i >= n else i)
    for i in range(cas, n, 2):
        x[i] -= (x[ext(i - 1)] + x[ext(i + 1)] + 2) >> 2
    for i in range(1 - cas, n, 2):
        x[i] += (x[ext(i - 1)] + x[ext(i + 1)]) >> 1
    return x


def _idwt53_level(f, x0, y0):
    h, w = f.shape
    out = f.astype(np.int64).copy()
    cx, cy = x0 & 1, y0 & 1
    sw, sh = (w + 1 - cx) // 2, (h + 1 - cy) // 2
    for y in range(h):
        out[y] = _inv53_line(out[y, :sw].copy(), out[y, sw:].copy(), cx)
    for x in range(w):
        out[:, x] = _inv53_line(out[:sh, x].copy(), out[sh:, x].copy(), cy)
    return out.astype(np.int32)


# ---- live library (skipped where no libopenjp2 is present) ---------------------------------------

LIVE = [
    (64, 64, 1, 8, "A", dict(numres=2)),
    (300, 200, 3, 8, "A", dict(numres=6, mct=True)),
    (300, 200, 3, 16, "A", dict(numres=6, mct=True, tile=128)),
    (300, 200, 3, 8, "A", dict(numres=6, mct=False, layers=12, tile=1024)),
    (300, 200, 4, 8, "B", dict(numres=6, mct=True, layers=3)),
    (300, 200, 3, 8, "A", dict(numres=6, mct=True, reversible=False)),
    (300, 200, 3, 16, "A", dict(numres=6, mct=True, reversible=False, tile=128)),
    (301, 199, 1, 10, "B", dict(numres=4)),
    (5, 3, 3, 12, "A", dict(numres=2, mct=True)),
    (128, 128, 1, 8, "B", dict(numres=6, cblk=(32, 32))),
    (130, 70, 1, 8, "B", dict(numres=3, cblk=(16, 64))),
    (100, 100, 1, 1, "A", dict(numres=3)),
]


@pytest.mark.parametrize("case", LIVE, ids=lambda c: f"{c[0]}x{c[1]}x{c[2]}p{c[3]}{c[4]}")
def test_oracle_bytes_equal_live_openjpeg(oracle, opj, case):
    w, h, nc, prec, dist, kw = case
    pl = synth.planes(w, h, nc, prec, 4711, dist)
    p = make_params(w, h, nc, prec, **kw)
    ref = opj.encode(pl, p)
    ours = oracle.encode(pl, p, comment=opj.comment)
    assert ours == ref
    if kw.get("reversible", True):
        assert np.array_equal(opj.decode(ours), pl)


def test_all_zero_and_constant_blocks(oracle, opj):
    for val in (0, 128, 255):
        pl = np.full((1, 70, 90), val, dtype=np.int32)
        p = make_params(90, 70, 1, 8, numres=3)
        assert oracle.encode(pl, p, comment=opj.comment) == opj.encode(pl, p)

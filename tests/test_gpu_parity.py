"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(include/j2k_hip.h), against the CPU oracle and the committed golden vectors.  Bit-exact for every
stage, both for the reversible 5/3 path and for the 9/7 path (every float product/sum is rounded
like the oracle's)."""
import hashlib
import os

import numpy as np
import pytest

from conftest import golden_case
from j2k_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def enc():
    from j2k_amd import api
    e = api.Encoder(0)
    yield e
    e.close()


def _api():
    from j2k_amd import api
    return api


# ------------------------------------------------------------------------------------------------ A1,A2,A4,A5
FRONT = [
    # (w, h, ncomp, prec, reversible, mct, row_pad, promote)
    (37, 11, 1, 8, True, False, 0, False),
    (64, 9, 3, 8, True, True, 0, False),
    (300, 20, 3, 8, False, True, 12, False),
    (129, 7, 3, 16, True, True, 8, False),
    (129, 7, 4, 16, False, True, 0, False),
    (70, 13, 3, 10, False, True, 0, False),
    (70, 13, 3, 12, True, False, 16, True),
]


@pytest.mark.parametrize("case", FRONT, ids=str)
def test_frontend_matches_oracle(enc, oracle, case):
    api = _api()
    w, h, nc, prec, rev, mct, pad, promote = case
    pl = synth.planes(w, h, nc, prec, 99)
    frame, lay = synth.ae_frame(pl, prec, row_pad_bytes=pad)
    p = api.make_params(w, h, nc, prec, reversible=rev, ycc=mct, promote=promote, num_resolutions=1)
    got = enc.stage_frontend(frame, lay, p)
    # oracle: CopyBuffer per channel (+ promote), then DC shift + colour transform
    sb = lay["sample_bytes"]
    src = frame
    if promote and sb == 2:
        v = frame.view(np.uint16).astype(np.uint32)
        src = np.where(v > 16384, ((v - 1) << 1) + 1, v << 1).astype(np.uint16).view(np.uint8)
    order = [lay["channel_offsets"][i] for i in (1, 2, 3, 0)]
    planes = [oracle.copy_channel(src, order[c], w, h, lay["colbytes"], lay["rowbytes"], sb, 8 * sb, prec)
              for c in range(nc)]
    ref = np.stack(planes).astype(np.int32)
    import ctypes as C
    ptrs = (C.POINTER(C.c_int32) * nc)(*[ref[c].ctypes.data_as(C.POINTER(C.c_int32)) for c in range(nc)])
    oracle.L.j2ko_dc_mct.argtypes = [C.POINTER(C.POINTER(C.c_int32)), C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int]
    oracle.L.j2ko_dc_mct(ptrs, nc, w * h, prec, int(rev), int(mct))
    assert np.array_equal(got.view(np.int32), ref)


UPSHIFT = [(70, 13, 3, 10, True, True), (70, 13, 3, 12, False, True), (33, 9, 1, 16, True, False), (64, 8, 4, 12, False, True)]


@pytest.mark.parametrize("case", UPSHIFT, ids=str)
def test_frontend_upshift_branch_matches_oracle(enc, oracle, case):
    """A2, the branch the After Effects layout never takes by itself: FileInfo.depth above the container depth
    (8-bit samples, target depth 10/12/16) -> CopyChannel's left shift with bit replication
    (src/common/j2k_codec.cpp:285-324).  Front end alone, then the whole codestream."""
    import ctypes as C
    api = _api()
    from oracle.oracle import make_params
    w, h, nc, prec, rev, mct = case
    pl8 = synth.planes(w, h, nc, 8, 4711)
    frame, lay = synth.ae_frame(pl8, 8, row_pad_bytes=4)
    assert lay["sample_bytes"] == 1
    p = api.make_params(w, h, nc, prec, reversible=rev, ycc=mct and nc >= 3, num_resolutions=3)
    got = enc.stage_frontend(frame, lay, p)
    order = [lay["channel_offsets"][i] for i in (1, 2, 3, 0)]
    up = np.stack([oracle.copy_channel(frame, order[c] if nc > 1 else order[0], w, h, lay["colbytes"], lay["rowbytes"], 1, 8, prec)
                   for c in range(nc)]).astype(np.int32)
    s = prec - 8
    assert np.array_equal(up, (pl8 << s) | (pl8 >> (8 - s)))  # bit replication, as the reference computes it
    ref = up.copy()
    ptrs = (C.POINTER(C.c_int32) * nc)(*[ref[c].ctypes.data_as(C.POINTER(C.c_int32)) for c in range(nc)])
    oracle.L.j2ko_dc_mct.argtypes = [C.POINTER(C.POINTER(C.c_int32)), C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int]
    oracle.L.j2ko_dc_mct(ptrs, nc, w * h, prec, int(rev), int(mct and nc >= 3))
    assert np.array_equal(got.view(np.int32), ref)
    cs = oracle.encode(up, make_params(w, h, nc, prec, reversible=rev, mct=mct and nc >= 3, numres=3))
    assert enc.encode_host(frame, lay, p) == cs
    d = enc.upload(frame)
    try:
        assert enc.encode_device(d, lay, p)[2] == cs
    finally:
        enc.free(d)


# ------------------------------------------------------------------------------------------------ A6
DWT = [(64, 64, 1, 0, 0), (300, 200, 5, 0, 0), (301, 199, 3, 0, 0), (128, 128, 5, 128, 128), (97, 61, 4, 33, 7),
       (1, 40, 2, 0, 0), (40, 1, 2, 1, 1), (2, 2, 1, 1, 0), (3, 5, 2, 0, 1), (1000, 37, 5, 0, 0), (513, 515, 6, 0, 0)]


@pytest.mark.parametrize("case", DWT, ids=str)
@pytest.mark.parametrize("rev", [True, False], ids=["53", "97"])
def test_dwt_matches_oracle_bit_exact(enc, oracle, case, rev):
    w, h, levels, x0, y0 = case
    rng = np.random.default_rng(w * 1000 + h)
    if rev:
        a = rng.integers(-40000, 40000, size=(2, h, w), dtype=np.int32)
        ref = np.stack([oracle.dwt53(a[i], levels, x0, y0) for i in range(2)])
    else:
        a = (rng.standard_normal((2, h, w)) * 3000).astype(np.float32)
        ref = np.stack([oracle.dwt97(a[i], levels, x0, y0) for i in range(2)])
    got, _ = enc.stage_dwt(a, levels, rev, x0, y0)
    assert np.array_equal(got.view(np.int32), ref.view(np.int32))


def test_dwt53_roundtrip_at_full_size(enc):
    """Size-independent property at BASELINE scale: one 5/3 level is inverted exactly by the
    textbook inverse (vectorised numpy), 8192x8192."""
    rng = np.random.default_rng(5)
    a = rng.integers(-32768, 32768, size=(1, 8192, 8192), dtype=np.int32)
    got, _ = enc.stage_dwt(a, 1, True)
    f = got[0].astype(np.int64)
    n = 8192
    sn = n // 2

    def inv(lo, hi):  # along last axis, even length, cas 0
        s = lo - ((np.concatenate([hi[..., :1], hi[..., :-1]], -1) + hi + 2) >> 2)
        d = hi + ((s + np.concatenate([s[..., 1:], s[..., -1:]], -1)) >> 1)
        out = np.empty(lo.shape[:-1] + (2 * lo.shape[-1],), dtype=np.int64)
        out[..., 0::2] = s
        out[..., 1::2] = d
        return out
    rows = inv(f[:, :sn], f[:, sn:])          # undo horizontal
    cols = inv(rows[:sn].T, rows[sn:].T).T    # undo vertical
    assert np.array_equal(cols, a[0])


# ------------------------------------------------------------------------------------------------ A7,A8
def _t1_cases(oracle, rev, prec, seed, dist):
    w, h = 200, 136
    pl = synth.planes(w, h, 1, prec, seed, dist)[0] - (1 << (prec - 1))
    if rev:
        coef = oracle.dwt53(pl, 2)
    else:
        coef = oracle.dwt97(pl.astype(np.float32), 2)
    # sub-band rectangles of the 2-level Mallat layout, cut into 64x64 (and partial) blocks
    rects, orients = [], []
    w1, h1, w2, h2 = (w + 1) // 2, (h + 1) // 2, (w + 3) // 4, (h + 3) // 4
    bands = [(0, 0, w2, h2, 0), (w2, 0, w1 - w2, h2, 1), (0, h2, w2, h1 - h2, 2), (w2, h2, w1 - w2, h1 - h2, 3),
             (w1, 0, w - w1, h1, 1), (0, h1, w1, h - h1, 2), (w1, h1, w - w1, h - h1, 3)]
    for (bx, by, bw, bh, o) in bands:
        for yy in range(0, bh, 64):
            for xx in range(0, bw, 64):
                rects.append((bx + xx, by + yy, min(64, bw - xx), min(64, bh - yy)))
                orients.append(o)
    return coef, rects, orients


@pytest.mark.parametrize("rev,prec,dist", [(True, 8, "A"), (True, 16, "A"), (True, 8, "B"), (False, 8, "A"),
                                           (False, 16, "A"), (False, 10, "B")])
def test_t1_blocks_match_oracle(enc, oracle, rev, prec, dist):
    coef, rects, orients = _t1_cases(oracle, rev, prec, 321, dist)
    step = 1.0 if rev else 0.37
    got = enc.stage_t1(coef, rects, orients, [step] * len(rects), rev)
    for r, o, g in zip(rects, orients, got):
        x, y, w, h = r
        blk = coef[y:y + h, x:x + w]
        if rev:
            data = (blk.astype(np.int64) << 6).astype(np.int32)
        else:
            data = np.array([[oracle.L.j2ko_quant97(float(v), step) for v in row] for row in blk], dtype=np.int32)
        ref = oracle.t1_block(data, o)
        assert g["numbps"] == ref["numbps"], (r, o)
        assert g["npasses"] == ref["npasses"], (r, o)
        assert g["length"] == len(ref["data"]), (r, o)
        assert g["data"] == ref["data"], (r, o)


@pytest.mark.parametrize("rev,prec,dist", [(True, 8, "A"), (False, 16, "A"), (False, 10, "B")])
def test_t1_pass_rates_and_distortion_match_oracle(enc, oracle, rev, prec, dist):
    """What rate control will consume: per-pass cumulative byte counts (after the reference's fix-ups)
    and per-pass distortion-LUT sums, produced by the DIST variant of the modelling kernel."""
    coef, rects, orients = _t1_cases(oracle, rev, prec, 99, dist)
    step = 1.0 if rev else 0.21
    got = enc.stage_t1(coef, rects, orients, [step] * len(rects), rev, want_passes=True)
    for r, o, g in zip(rects, orients, got):
        x, y, w, h = r
        blk = coef[y:y + h, x:x + w]
        if rev:
            data = (blk.astype(np.int64) << 6).astype(np.int32)
        else:
            data = np.array([[oracle.L.j2ko_quant97(float(v), step) for v in row] for row in blk], dtype=np.int32)
        ref = oracle.t1_block(data, o)
        assert g["data"] == ref["data"], (r, o)
        assert g["rates"] == ref["rates"], (r, o)
        assert g["nmsedec"] == ref["nmsedec"], (r, o)


def test_t1_degenerate_blocks(enc, oracle):
    """all-zero block, single non-zero sample, 1-wide and 1-high blocks, 4x4 block."""
    coef = np.zeros((64, 256), dtype=np.int32)
    coef[5, 70] = -3
    coef[:, 130] = np.arange(64) - 31
    coef[0, 192:256] = 7
    coef[10:14, 200:204] = np.arange(16).reshape(4, 4) - 8
    coef[20:27, 140:153] = (np.arange(91).reshape(7, 13) % 23) - 11
    # (rectangles must not overlap: the kernel rewrites each block in place as scaled magnitudes)
    rects = [(0, 0, 64, 64), (64, 0, 64, 64), (130, 0, 1, 64), (192, 0, 64, 1), (200, 10, 4, 4), (140, 20, 13, 7)]
    orients = [0, 3, 1, 2, 0, 3]
    got = enc.stage_t1(coef, rects, orients, [1.0] * len(rects), True)
    for r, o, g in zip(rects, orients, got):
        x, y, w, h = r
        ref = oracle.t1_block((coef[y:y + h, x:x + w].astype(np.int64) << 6).astype(np.int32), o)
        assert (g["numbps"], g["npasses"], g["data"]) == (ref["numbps"], ref["npasses"], ref["data"]), r


# ------------------------------------------------------------------------------------------------ whole path
SMALL = ["g1_64x64_grey_1lvl", "g1_64x64_grey_5lvl", "g2_c1_512_grey_53", "g3_300x200_rgb8_53_rct",
         "g4_300x200_rgb16_53_rct_tile128", "g5_300x200_rgb8_53_ref_literal", "g6_300x200_rgb8_97_ict",
         "g6_300x200_rgb16_97_ict", "g7_300x200_rgb10_53", "g9_300x200_rgba8_53_rct",
         "g9_97x61_grey12_97_4lvl", "g9_150x130_rgb8_97_tile64"]


def _params_from_golden(g):
    api = _api()
    kw = g["params"]
    return api.make_params(g["width"], g["height"], g["ncomp"], g["prec"], reversible=kw.get("reversible", True),
                           ycc=kw.get("mct", False), layers=kw.get("layers", 1), tile_size=kw.get("tile", 0),
                           num_resolutions=kw.get("numres", 6), cblk=tuple(kw.get("cblk", (64, 64))), comment="")


@pytest.mark.parametrize("name", SMALL)
def test_codestream_equals_golden(enc, golden, name):
    g, pl, _, cs = golden_case(golden, name)
    frame, lay = synth.ae_frame(pl, g["prec"], row_pad_bytes=8)
    p = _params_from_golden(g)
    ours = enc.encode_host(frame, lay, p)
    assert len(ours) == g["length"]
    assert ours == cs
    # same through the sink callback and from a device-resident frame
    assert enc.encode_host(frame, lay, p, via_sink=True) == cs
    d = enc.upload(frame)
    try:
        assert enc.encode_device(d, lay, p)[2] == cs
    finally:
        enc.free(d)


@pytest.mark.parametrize("name", SMALL)
def test_fused_level1_in_workgroups_of_four_small_goldens(enc, golden, name):
    """The fused level-1 kernel's big-frame form -- workgroups of four adjacent strips that meet at a barrier after every row
    pair -- forced onto the small goldens: strips beyond the image (waves that leave at once), edge strips beside fast ones in
    one workgroup, odd origins of tiles, single rows; and the generic sample extraction beside the compile-time one."""
    api = _api()
    g, pl, _, cs = golden_case(golden, name)
    frame, lay = synth.ae_frame(pl, g["prec"], row_pad_bytes=8)
    p = _params_from_golden(g)
    try:
        for wpb, generic in ((4, 0), (4, 1), (1, 1)):
            api.tune("fused_wpb", wpb)
            api.tune("fused_generic", generic)
            assert enc.encode_host(frame, lay, p) == cs, (wpb, generic)
    finally:
        api.tune("fused_wpb", 0)
        api.tune("fused_generic", 0)


def test_fused_occupancy_is_read_from_the_code_object(enc):
    """the chunk heuristic of launch_fused sizes a "round" of resident waves by the kernels' register counts
    -- now read from the built code objects, not written down.  What DESIGN.md quotes for today's build: 9/7 RGB three waves
    per SIMD; none of the variants may fall to one."""
    occ = {(rev, nc): enc.L.j2k_hip_debug_fused_occupancy(enc.h, int(rev), nc) for rev in (False, True) for nc in (1, 3, 4)}
    assert all(2 <= v <= 8 for v in occ.values()), occ
    assert occ[(False, 3)] == 3, occ


def test_codestream_equals_oracle_random_shapes(enc, oracle):
    api = _api()
    rng = np.random.default_rng(2024)
    for i in range(10):
        w, h = int(rng.integers(33, 400)), int(rng.integers(33, 300))
        nc = int(rng.choice([1, 3, 4]))
        prec = int(rng.choice([8, 10, 12, 16]))
        rev = bool(rng.integers(0, 2))
        mct = nc >= 3 and bool(rng.integers(0, 2))
        numres = int(rng.integers(1, 6))
        tile = int(rng.choice([0, 0, 64, 100, 128]))
        if tile and tile < (1 << (numres - 1)):
            tile = 0
        from oracle.oracle import make_params
        pl = synth.planes(w, h, nc, prec, 1000 + i, "A" if i % 2 else "B")
        ref = oracle.encode(pl, make_params(w, h, nc, prec, reversible=rev, mct=mct, numres=numres, tile=tile))
        frame, lay = synth.ae_frame(pl, prec)
        p = api.make_params(w, h, nc, prec, reversible=rev, ycc=mct, num_resolutions=numres, tile_size=tile)
        assert enc.encode_host(frame, lay, p) == ref, (w, h, nc, prec, rev, mct, numres, tile)


def test_tile_sharded_encode_concatenates_to_full(enc, golden):
    """tile-parts produced for disjoint tile ranges + main header + EOC == full codestream."""
    api = _api()
    import ctypes as C
    g, pl, _, cs = golden_case(golden, "g4_300x200_rgb16_53_rct_tile128")
    frame, lay = synth.ae_frame(pl, g["prec"])
    p = _params_from_golden(g)
    hdr = np.empty(1024, dtype=np.uint8)
    n, nt = C.c_size_t(), C.c_uint32()
    assert enc.L.j2k_hip_main_header(C.byref(p), hdr.ctypes.data, 1024, C.byref(n), C.byref(nt)) == 0
    assert nt.value == 6
    d = enc.upload(frame)
    try:
        parts = [enc.encode_tiles_device(d, lay, p, a, b) for (a, b) in [(0, 1), (1, 3), (4, 2)]]
    finally:
        enc.free(d)
    assert hdr[:n.value].tobytes() + b"".join(parts) + b"\xff\xd9" == cs


def test_error_reporting(enc):
    api = _api()
    pl = synth.planes(64, 64, 1, 8, 1)
    frame, lay = synth.ae_frame(pl, 8)
    with pytest.raises(api.J2kHipError) as ei:
        enc.encode_host(frame, lay, api.make_params(64, 64, 1, 8, num_resolutions=8))
    assert "Number of resolutions is too high" in str(ei.value)
    with pytest.raises(api.J2kHipError):
        enc.encode_host(frame, lay, api.make_params(64, 64, 1, 8, ycc=True))


FULL = ["c1_512_grey_53", "c5_frame0_4096x2160_rgb10_97", "c4_tile_2048_rgb16_53", "c2_4096_rgb8_97",
        "c3_8192_rgb16_97_5lvl", "c3_8192_rgb16_97_6lvl"]


@pytest.mark.parametrize("name", FULL)
def test_full_size_codestream_hash(enc, golden, name):
    """BASELINE-size configs: the codestream hash equals the one libopenjp2 produced here."""
    if name not in golden:
        pytest.skip("full-size golden not generated")
    g, pl, _, _ = golden_case(golden, name)
    frame, lay = synth.ae_frame(pl, g["prec"])
    del pl
    p = _params_from_golden(g)
    ours = enc.encode_host(frame, lay, p)
    assert len(ours) == g["length"]
    assert hashlib.sha256(ours).hexdigest() == g["sha256"]


def test_c4_tile_row_sharded_equals_libopenjp2(enc, golden):
    """BASELINE config 4 as one rank of the 8-GPU job sees it: a 16384 x 2048 slice = 8 tiles of 2048^2 (all but
    the first at a non-zero origin), 16-bit RGB, 5/3 + RCT, encoded as two tile ranges and put together like
    rank 0 does (main header + tile-parts + EOC)."""
    import ctypes as C
    name = "c4_slice_16384x2048_rgb16_53_tile2048"
    if name not in golden:
        pytest.skip("full-size golden not generated")
    g, pl, _, _ = golden_case(golden, name)
    frame, lay = synth.ae_frame(pl, g["prec"])
    del pl
    p = _params_from_golden(g)
    hdr = np.empty(1024, dtype=np.uint8)
    n, nt = C.c_size_t(), C.c_uint32()
    assert enc.L.j2k_hip_main_header(C.byref(p), hdr.ctypes.data, 1024, C.byref(n), C.byref(nt)) == 0
    assert nt.value == 8
    d = enc.upload(frame)
    try:
        parts = [enc.encode_tiles_device(d, lay, p, a, b) for (a, b) in [(0, 3), (3, 5)]]
        whole = enc.encode_device(d, lay, p)[2]
    finally:
        enc.free(d)
    cs = hdr[:n.value].tobytes() + b"".join(parts) + b"\xff\xd9"
    assert len(cs) == g["length"]
    assert hashlib.sha256(cs).hexdigest() == g["sha256"]
    assert whole == cs


def test_timed_configuration_is_byte_exact(golden):
    """The configuration bench.py times -- several handles in flight on the metric frame (8192^2 RGB16 9/7, 5
    levels: 49,152 blocks, so two coder groups, the hand-shake that holds the bulk coder launch back behind the
    next frame's DWT, coder waves that yield to it) -- under a byte check.  (The scalar coder of the longest
    streams runs when a frame is alone on the device: the single-handle full-size cases.)"""
    import ctypes as C
    import threading
    api = _api()
    name = "c3_8192_rgb16_97_5lvl"
    g, pl, _, _ = golden_case(golden, name)
    frame, lay = synth.ae_frame(pl, g["prec"])
    del pl
    p = _params_from_golden(g)
    up = api.Encoder(0)
    d = up.upload(frame)
    del frame
    hashes, errors = [], []

    def worker(k):
        e = api.Encoder(0)
        try:
            for _ in range(3):
                dptr, n, _ = e.encode_device(d, lay, p, download=False)
                hashes.append(hashlib.sha256(e.d2h(dptr, n)).hexdigest())
        except Exception as ex:  # noqa: BLE001
            errors.append(repr(ex))
        finally:
            e.close()

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    up.free(d)
    up.close()
    assert not errors, errors
    assert len(hashes) == 9 and set(hashes) == {g["sha256"]}


KNOBS = [("mq_yield", 0), ("mq_yield", 1), ("dwt_ahead", 1), ("overlap", 0), ("mq_single", 1), ("heavy_min", 30000),
         ("groups", 3), ("mq_wait_us", 0), ("dense_chain", 0), ("level1_dispatch_events", 0), ("fused_wpb", 1), ("fused_generic", 1)]


@pytest.mark.parametrize("knob,value", KNOBS)
def test_tuning_knobs_never_change_a_byte(golden, knob, value):
    """Every scheduling / variant knob of the library (coder yield, DWT ahead of the previous modeller, no overlap,
    the one-wave coder, a low scalar-coder threshold, three coder groups, no hold-back) with two handles in flight
    on the metric frame: the codestream stays libopenjp2's."""
    import threading
    api = _api()
    name = "c3_8192_rgb16_97_5lvl"
    g, pl, _, _ = golden_case(golden, name)
    frame, lay = synth.ae_frame(pl, g["prec"])
    del pl
    p = _params_from_golden(g)
    up = api.Encoder(0)
    d = up.upload(frame)
    del frame
    defaults = {"mq_yield": 2, "dwt_ahead": 0, "overlap": 1, "mq_single": 0, "heavy_min": 0, "groups": 2, "mq_wait_us": 1500, "dense_chain": 1,
                "level1_dispatch_events": 1, "fused_wpb": 0, "fused_generic": 0}
    hashes, errors = [], []

    def worker():
        e = api.Encoder(0)
        try:
            for _ in range(2):
                dptr, n, _ = e.encode_device(d, lay, p, download=False)
                hashes.append(hashlib.sha256(e.d2h(dptr, n)).hexdigest())
        except Exception as ex:  # noqa: BLE001
            errors.append(repr(ex))
        finally:
            e.close()

    api.tune(knob, value)
    try:
        ths = [threading.Thread(target=worker) for _ in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
    finally:
        api.tune(knob, defaults[knob])
        up.free(d)
        up.close()
    assert not errors, errors
    assert len(hashes) == 4 and set(hashes) == {g["sha256"]}


# ------------------------------------------------------------------------------------------------ C++ codec interface
def _host_write(frame, lay, w, h, channels, depth, reversible, ycc, layers, tile, honour, max_write=-1):
    import ctypes as C
    from j2k_amd import api
    api.load_library()
    H = C.CDLL(os.path.join(os.path.dirname(api.LIBPATH), "libj2k_host.so"))
    H.j2k_host_test_write.restype = C.c_long
    H.j2k_host_test_write.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_ulong, C.c_char_p, C.c_ulong]
    out = np.empty(frame.nbytes * 2 + (1 << 20), dtype=np.uint8)
    err = C.create_string_buffer(512)
    n = H.j2k_host_test_write(frame.ctypes.data, w, h, lay["rowbytes"], lay["sample_bytes"], channels, depth, int(reversible),
                              int(ycc), layers, tile, int(honour), max_write, out.ctypes.data, out.nbytes, err, 512)
    return (out[:n].tobytes() if n >= 0 else None), err.value.decode()


def test_hip_codec_reference_literal_matches_golden(golden):
    """HipCodec::WriteFile driven like the plug-in drives a Codec (WorldToBuffer layout, channelMap,
    base-class call): the reference's literal parameterisation (5/3, MCT off, 12 layers, tile 1024)."""
    from oracle.oracle import strip_com
    g, pl, _, cs = golden_case(golden, "g5_300x200_rgb8_53_ref_literal")
    frame, lay = synth.ae_frame(pl, 8, row_pad_bytes=4)
    # settings.reversible/ycc are ignored in this mode, exactly like the reference adapter ignores them
    got, err = _host_write(frame, lay, 300, 200, 3, 8, reversible=False, ycc=True, layers=12, tile=1024, honour=False)
    assert got is not None, err
    assert strip_com(got) == cs
    assert b"Created by j2k_hip" in got[:200]


def test_hip_codec_honour_settings_and_errors(golden):
    from oracle.oracle import strip_com
    g, pl, _, cs = golden_case(golden, "g6_300x200_rgb16_97_ict")
    frame, lay = synth.ae_frame(pl, 16)
    got, err = _host_write(frame, lay, 300, 200, 3, 16, reversible=False, ycc=True, layers=1, tile=0, honour=True)
    assert got is not None, err
    assert strip_com(got) == cs
    g, pl, _, cs = golden_case(golden, "g9_300x200_rgba8_53_rct")
    frame, lay = synth.ae_frame(pl, 8)
    got, err = _host_write(frame, lay, 300, 200, 4, 8, reversible=True, ycc=True, layers=1, tile=0, honour=True)
    assert got is not None and strip_com(got) == cs
    # error convention: any failure surfaces as j2k::Exception("Error writing file")
    got, err = _host_write(frame, lay, 300, 200, 4, 8, True, True, 1, 0, True, max_write=100)  # short write in the sink
    assert got is None and err.startswith("Error writing file")
    got, err = _host_write(frame, lay, 300, 200, 4, 8, True, True, 1, 16, True)  # tile too small for 6 resolutions
    assert got is None and err.startswith("Error writing file") and "resolutions" in err


@pytest.mark.gpu
def test_concurrent_handles_overlap_and_stay_exact(golden):
    """Frames in flight: several host threads, one encoder handle each, different images and coding
    parameters at the same time (their GPU phases overlap by design) -- every codestream must still be
    the golden one, every time."""
    import threading
    api = _api()
    names = ["g3_300x200_rgb8_53_rct", "g6_300x200_rgb16_97_ict", "g4_300x200_rgb16_53_rct_tile128",
             "g9_150x130_rgb8_97_tile64"]
    jobs = []
    for name in names:
        g, pl, _, cs = golden_case(golden, name)
        frame, lay = synth.ae_frame(pl, g["prec"])
        kw = g["params"]
        p = api.make_params(g["width"], g["height"], g["ncomp"], g["prec"], reversible=kw.get("reversible", True),
                            ycc=kw.get("mct", False), tile_size=kw.get("tile", 0), num_resolutions=kw.get("numres", 6))
        jobs.append((frame, lay, p, cs))
    # one big frame keeps coder chains on the chip while the small ones run
    big = synth.planes(2048, 2048, 3, 16, 77, "A")
    bframe, blay = synth.ae_frame(big, 16)
    bp = api.make_params(2048, 2048, 3, 16, reversible=False, ycc=True)
    errors, big_out = [], []

    def small(k):
        enc = api.Encoder(0)
        try:
            for it in range(6):
                frame, lay, p, cs = jobs[(k + it) % len(jobs)]
                if enc.encode_host(frame, lay, p, via_sink=bool(it & 1)) != cs:
                    errors.append((k, it))
        finally:
            enc.close()

    def large():
        enc = api.Encoder(0)
        try:
            for _ in range(3):
                big_out.append(enc.encode_host(bframe, blay, bp))
        finally:
            enc.close()

    ths = [threading.Thread(target=small, args=(k,)) for k in range(3)] + [threading.Thread(target=large)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors
    assert len(big_out) == 3 and big_out[0] == big_out[1] == big_out[2]
    enc = api.Encoder(0)
    assert enc.encode_host(bframe, blay, bp) == big_out[0]  # same bytes with the chip to itself
    enc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,extra", [("g3_300x200_rgb8_53_rct", {}), ("g6_300x200_rgb16_97_ict", {}),
                                        ("g4_300x200_rgb16_53_rct_tile128", {}), ("g9_300x200_rgba8_53_rct", {}),
                                        ("g6_300x200_rgb8_97_ict", dict(rates=[30.0, 8.0])),
                                        ("g3_300x200_rgb8_53_rct", dict(jp2=True, color_space=1))])
def test_frame_sequence_equals_frame_by_frame(golden, name, extra):
    """j2k_hip_encode_sequence_device: the frames of a sequence share the Tier-1 launches; every codestream must be
    the one the frame gets on its own (frame 0 is the golden image, the others differ in content)."""
    api = _api()
    g, pl0, _, cs = golden_case(golden, name)
    kw = g["params"]
    p = api.make_params(g["width"], g["height"], g["ncomp"], g["prec"], reversible=kw.get("reversible", True),
                        ycc=kw.get("mct", False), tile_size=kw.get("tile", 0), num_resolutions=kw.get("numres", 6), **extra)
    frames = [pl0] + [synth.planes(g["width"], g["height"], g["ncomp"], g["prec"], 900 + k, "AB"[k & 1]) for k in range(4)]
    enc = api.Encoder(0)
    lay = None
    dptrs, singles = [], []
    for pl in frames:
        fr, lay = synth.ae_frame(pl, g["prec"])
        d = enc.upload(fr)
        dptrs.append(d)
        singles.append(enc.encode_device(d, lay, p)[2])
    if not extra:
        assert singles[0] == cs
    seq = enc.encode_sequence_device(dptrs, lay, p)
    assert [x[2] for x in seq] == singles
    # a different length right after, and a single frame again: cached tables must follow
    seq2 = enc.encode_sequence_device(dptrs[1:3], lay, p)
    assert [x[2] for x in seq2] == singles[1:3]
    assert enc.encode_device(dptrs[4], lay, p)[2] == singles[4]
    for d in dptrs:
        enc.free(d)
    enc.close()


def test_codestream_equals_oracle_random_block_sizes(enc, oracle):
    """Code-block sizes other than 64 x 64 (16 / 32 / 64 per side, mixed), with and without byte budgets, widths that
    leave partial blocks: the modeller keeps its bit-planes transposed for 64- and 32-row blocks and reads the rows
    of every other height in place; with budgets it also sums the distortion estimates from those planes."""
    api = _api()
    from oracle.oracle import make_params
    rng = np.random.default_rng(int(os.environ.get("J2K_FUZZ_SEED", "31337")))
    for i in range(int(os.environ.get("J2K_FUZZ_CASES", "14"))):
        big = int(os.environ.get("J2K_FUZZ_MAX", "420"))  # (larger by hand: frames with whole 64 x 64 blocks inside)
        w, h = int(rng.integers(40, big)), int(rng.integers(40, max(41, big * 3 // 4)))
        nc = int(rng.choice([1, 3]))
        prec = int(rng.choice([8, 10, 12, 16]))
        rev = bool(rng.integers(0, 2))
        numres = int(rng.integers(1, 5))
        cb = (int(rng.choice([16, 32, 64])), int(rng.choice([16, 32, 64])))
        rates = [None, None, [25.0], [40.0, 12.0], [30.0, 10.0, 0.0]][int(rng.integers(0, 5))]
        prog = int(rng.integers(0, 5))
        pl = synth.planes(w, h, nc, prec, 5000 + i, "A" if i % 2 else "B")
        op = make_params(w, h, nc, prec, reversible=rev, mct=nc == 3, numres=numres, cblk=cb, prog=prog, layers=len(rates) if rates else 1)
        ref = oracle.encode_rates(pl, op, rates) if rates else oracle.encode(pl, op)
        frame, lay = synth.ae_frame(pl, prec)
        p = api.make_params(w, h, nc, prec, reversible=rev, ycc=nc == 3, num_resolutions=numres, cblk=cb, rates=rates, progression=prog)
        assert enc.encode_host(frame, lay, p) == ref, (w, h, nc, prec, rev, numres, cb, rates, prog)
        if rates:  # the allocation's per-block work on the device (rate.hip), every scan or the big rounds only
            try:
                api.tune("rate_dev", 1)
                for scan in (1, 40):
                    api.tune("rate_dev_scan", scan)
                    assert enc.encode_host(frame, lay, p) == ref, ("rate_dev", scan, w, h, nc, prec, rev, numres, cb, rates, prog)
            finally:
                api.tune("rate_dev", 0)
                api.tune("rate_dev_scan", 0)


@pytest.mark.parametrize("heavy_min", [72000, 30000])
def test_scalar_coder_for_long_streams_never_changes_a_byte(golden, heavy_min):
    """The wave-per-block scalar coder (off by default: `heavy_min` = 0) takes the blocks with at least `heavy_min`
    decisions when a frame is alone on the device: one handle, the metric frame, libopenjp2's hash."""
    api = _api()
    name = "c3_8192_rgb16_97_5lvl"
    g, pl, _, _ = golden_case(golden, name)
    frame, lay = synth.ae_frame(pl, g["prec"])
    del pl
    p = _params_from_golden(g)
    e = api.Encoder(0)
    d = e.upload(frame)
    del frame
    api.tune("heavy_min", heavy_min)
    try:
        for _ in range(2):
            dptr, n, _ = e.encode_device(d, lay, p, download=False)
            assert hashlib.sha256(e.d2h(dptr, n)).hexdigest() == g["sha256"]
    finally:
        api.tune("heavy_min", 0)
        e.free(d)
        e.close()


# ------------------------------------------------------------------------------------------------ entry points of the plug-in and the bench
@pytest.mark.parametrize("staging", [0, 1], ids=["direct", "staged"])
def test_pipelined_begin_end_is_byte_exact(golden, staging):
    """j2k_hip_encode_begin / _end -- the API INTEGRATION.md recommends for an image sequence and bench.py's host_path times:
    one host thread, three handles, begin(h0,f0) begin(h1,f1) end(h0) begin(h2,f2) ... over small goldens and the 4096^2
    frame (which is large enough for the staged upload, two coder groups and the pieced download).  Also with the pinned
    double-buffered staging of the upload (`staging` = 1)."""
    api = _api()
    jobs = []
    for name in ("g6_300x200_rgb16_97_ict", "g4_300x200_rgb16_53_rct_tile128", "c2_4096_rgb8_97", "g6_300x200_rgb16_97_ict",
                 "c2_4096_rgb8_97", "g4_300x200_rgb16_53_rct_tile128", "c2_4096_rgb8_97"):
        g, pl, _, cs = golden_case(golden, name)
        frame, lay = synth.ae_frame(pl, g["prec"], row_pad_bytes=0 if name.startswith("c2") else 8)
        jobs.append((name, frame, lay, _params_from_golden(g), cs, g))
    encs = [api.Encoder(0) for _ in range(3)]
    api.tune("staging", staging)
    api.tune("stage_kb", 4096)
    got = {}
    try:
        n = len(jobs)
        for i in range(n + 2):
            if i >= 2:
                got[i - 2] = encs[(i - 2) % 3].encode_end()
            if i < n:
                _, frame, lay, p, _, _ = jobs[i]
                encs[i % 3].encode_begin_host(frame, lay, p)
                frame[:] = 0xEE  # the caller's buffer may be reused at once: _begin has taken the frame
    finally:
        api.tune("staging", 0)
        api.tune("stage_kb", 16384)
        for e in encs:
            e.close()
    for i, (name, _, _, _, cs, g) in enumerate(jobs):
        if cs is not None:
            assert got[i] == cs, (i, name)
        else:
            assert len(got[i]) == g["length"] and hashlib.sha256(got[i]).hexdigest() == g["sha256"], (i, name)


def test_borrowed_begin_is_byte_exact_and_guards_the_handle(golden):
    """j2k_hip_encode_begin_borrowed: the deferred _begin (upload + launches on a thread of the handle) over three handles
    from one thread, frames left alone until _end; byte-identical files.  Between the two calls the handle refuses
    everything else; a failure of the deferred half (a plane pointer that cannot be read is not testable without a crash:
    a row stride the frame cannot have) surfaces at _end; destroying a handle with a deferred begin on its way is safe."""
    api = _api()
    jobs = []
    for name in ("c2_4096_rgb8_97", "g6_300x200_rgb16_97_ict", "g4_300x200_rgb16_53_rct_tile128", "c2_4096_rgb8_97", "g6_300x200_rgb16_97_ict"):
        g, pl, _, cs = golden_case(golden, name)
        frame, lay = synth.ae_frame(pl, g["prec"], row_pad_bytes=0 if name.startswith("c2") else 8)
        jobs.append((name, frame, lay, _params_from_golden(g), cs, g))
    encs = [api.Encoder(0) for _ in range(3)]
    got = {}
    try:
        n = len(jobs)
        for i in range(n + 2):
            if i >= 2:
                got[i - 2] = encs[(i - 2) % 3].encode_end()
            if i < n:
                _, frame, lay, p, _, _ = jobs[i]
                encs[i % 3].encode_begin_borrowed(frame, lay, p)
        for i, (name, _, _, _, cs, g) in enumerate(jobs):
            if cs is not None:
                assert got[i] == cs, (i, name)
            else:
                assert len(got[i]) == g["length"] and hashlib.sha256(got[i]).hexdigest() == g["sha256"], (i, name)
        # the handle is busy between the two calls
        name, frame, lay, p, cs, g = jobs[1]
        e = encs[0]
        e.encode_begin_borrowed(frame, lay, p)
        for call in (lambda: e.encode_begin_borrowed(frame, lay, p), lambda: e.encode_begin_host(frame, lay, p), lambda: e.encode_host(frame, lay, p),
                     lambda: e.decode_planar(cs)):
            with pytest.raises(api.J2kHipError, match="in progress"):
                call()
        assert e.encode_end() == cs   # ... and none of the refused calls disturbed the frame on its way
        # parameters that do not normalise are refused by the call itself, not by _end
        bad = _params_from_golden(g)
        bad.cblk_w = 48
        with pytest.raises(api.J2kHipError):
            e.encode_begin_borrowed(frame, lay, bad)
        assert e.encode_host(frame, lay, p) == cs
        # a handle destroyed with its deferred begin on the way
        e2 = api.Encoder(0)
        e2.encode_begin_borrowed(jobs[0][1], jobs[0][2], jobs[0][3])
        e2.close()
    finally:
        for e in encs:
            e.close()


# ------------------------------------------------------------------------------------------------ band-pipelined host calls (bands.h)
def _with_bands(n):
    import contextlib
    api = _api()

    @contextlib.contextmanager
    def ctx():
        api.tune("bands", n)
        try:
            yield
        finally:
            api.tune("bands", 0)
    return ctx()


@pytest.mark.parametrize("bands", [1, 4, 8])
@pytest.mark.parametrize("name", ["c2_4096_rgb8_97", "g4_300x200_rgb16_53_rct_tile128", "c3_8192_rgb16_97_5lvl"])
def test_band_pipelined_encode_is_byte_exact(enc, golden, name, bands):
    """the synchronous host call (what HipCodec::WriteFile makes) with the frame uploaded in 1, 4 and 8
    row bands -- level 1 of the DWT and the Tier-1 of finished bands running beside the upload, finished stages coming down
    beside the coding of later ones, the file handed to the sink in pieces -- against libopenjp2's files: the 4096^2
    frame, the tiled odd-sized one and the metric frame; into a buffer and through the sink."""
    if name not in golden:
        pytest.skip("full-size golden not generated")
    g, pl, _, cs = golden_case(golden, name)
    frame, lay = synth.ae_frame(pl, g["prec"], row_pad_bytes=0 if name.startswith("c") else 8)
    del pl
    p = _params_from_golden(g)
    with _with_bands(bands):
        a = enc.encode_host(frame, lay, p)
        st = enc.stats()
        b = enc.encode_host(frame, lay, p, via_sink=True)
    for ours in (a, b):
        assert len(ours) == g["length"]
        assert (ours == cs) if cs is not None else (hashlib.sha256(ours).hexdigest() == g["sha256"])
    from j2k_amd import api as _a  # (the schedule's own count: small images have fewer bands than asked for)
    assert 1 <= st["bands"] <= bands and st["ms_after_upload"] > 0
    if bands > 1 and g["height"] >= 4096:
        assert st["bands"] == bands and st["early_download_bytes"] > g["length"] // 3, st


@pytest.mark.parametrize("name", SMALL)
def test_band_pipelined_small_goldens(enc, golden, name):
    """Every small golden (grey / RGB / RGBA, 5/3 and 9/7, one and five levels, tiles of 64 and 128, 8..16 bits) through the
    band machinery with three bands forced: images lower than a band, tiles that end inside a band, single-level frames."""
    g, pl, _, cs = golden_case(golden, name)
    frame, lay = synth.ae_frame(pl, g["prec"], row_pad_bytes=8)
    p = _params_from_golden(g)
    with _with_bands(3):
        assert enc.encode_host(frame, lay, p) == cs
        banded = enc.stats()["bands"]
        assert enc.encode_host(frame, lay, p, via_sink=True) == cs
    if g["params"].get("numres", 6) > 1:
        assert banded >= 1, "the call did not take the band-pipelined path"


def test_band_pipelined_random_shapes_promote_and_jp2(enc, oracle):
    """Random sizes, tiles, depths, channel counts (alpha), Promote and the JP2 wrapper with 2..5 bands forced, against the
    oracle -- and the begin/_end and borrowed forms of the call, which run the same two halves."""
    api = _api()
    from oracle.oracle import make_params
    rng = np.random.default_rng(77)
    for i in range(12):
        w, h = int(rng.integers(40, 700)), int(rng.integers(140, 900))
        nc = int(rng.choice([1, 3, 4]))
        prec = int(rng.choice([8, 10, 12, 16]))
        rev = bool(rng.integers(0, 2))
        mct = nc >= 3 and bool(rng.integers(0, 2))
        numres = int(rng.integers(2, 6))
        tile = int(rng.choice([0, 0, 100, 128, 256]))
        promote = prec == 16 and bool(rng.integers(0, 2))
        pl = synth.planes(w, h, nc, prec, 3000 + i, "A" if i % 2 else "B")
        src = pl
        if promote:  # the host's world holds 15+1-bit samples: what Promote() makes of them is what gets coded
            src = np.minimum(pl.astype(np.int64) >> 1, 32768).astype(pl.dtype)
            v = src.astype(np.int64)
            pl = np.where(v > 16384, ((v - 1) << 1) + 1, v << 1).astype(pl.dtype)
        ref = oracle.encode(pl, make_params(w, h, nc, prec, reversible=rev, mct=mct, numres=numres, tile=tile))
        frame, lay = synth.ae_frame(src, prec, row_pad_bytes=int(rng.choice([0, 8, 24])))
        p = api.make_params(w, h, nc, prec, reversible=rev, ycc=mct, num_resolutions=numres, tile_size=tile, promote=promote)
        with _with_bands(int(rng.integers(2, 6))):
            got = enc.encode_host(frame, lay, p)
            assert enc.stats()["bands"] >= 1
            assert got == ref, (i, w, h, nc, prec, rev, mct, numres, tile, promote)
            if i % 3 == 0:
                enc.encode_begin_host(frame, lay, p)
                keep = frame.copy()
                frame[:] = 0x5A
                assert enc.encode_end() == ref, ("begin/end", i)
                frame[:] = keep
            if i % 3 == 1:
                enc.encode_begin_borrowed(frame, lay, p)
                assert enc.encode_end() == ref, ("borrowed", i)
    # the JP2 wrapper: file header pieces in front of the codestream's
    w, h = 333, 517
    pl = synth.planes(w, h, 3, 8, 5)
    frame, lay = synth.ae_frame(pl, 8)
    p = api.make_params(w, h, 3, 8, reversible=False, ycc=True, num_resolutions=5, jp2=True, color_space=1)
    plain = enc.encode_host(frame, lay, p)
    with _with_bands(4):
        assert enc.encode_host(frame, lay, p) == plain and enc.stats()["bands"] >= 2
    # an output buffer that is too small reports the length it needs
    import ctypes as C
    planes = api.planes_from_layout(frame.ctypes.data, lay, 3)
    n = C.c_size_t()
    small = np.empty(100, dtype=np.uint8)
    with _with_bands(4):
        rc = enc.L.j2k_hip_encode_to_buffer(enc.h, C.byref(p), planes, small.ctypes.data, small.nbytes, C.byref(n))
    assert rc == 4 and n.value == len(plain)


def test_c4_whole_64_tile_image_through_the_host_call(golden):
    """BASELINE config 4 whole: 16384 x 16384 16-bit RGB, 64 tiles of 2048^2, 5/3 + RCT, from a host frame
    through the synchronous call -- band-pipelined: every band is a row of eight tiles, whose five levels and code-blocks are
    done while the next tile row is on its way -- against libopenjp2's hash of the whole 828 MB codestream."""
    name = "c4_16384_rgb16_53_tile2048"
    if name not in golden:
        pytest.skip("full-size golden not generated")
    api = _api()
    g, pl, _, _ = golden_case(golden, name)
    frame, lay = synth.ae_frame(pl, g["prec"])
    del pl
    p = _params_from_golden(g)
    e = api.Encoder(0)
    try:
        with _with_bands(8):
            cs = e.encode_host(frame, lay, p)
            st = e.stats()
        assert len(cs) == g["length"] and hashlib.sha256(cs).hexdigest() == g["sha256"]
        assert st["bands"] == 8 and st["early_download_bytes"] > g["length"] // 2, st
        del cs
        api.tune("bands", -1)  # and in one piece
        try:
            cs = e.encode_host(frame, lay, p)
        finally:
            api.tune("bands", 0)
        assert hashlib.sha256(cs).hexdigest() == g["sha256"]
    finally:
        e.close()


def test_begin_without_end_and_end_without_begin_fail_cleanly(enc, golden):
    api = _api()
    g, pl, _, cs = golden_case(golden, "g3_300x200_rgb8_53_rct")
    frame, lay = synth.ae_frame(pl, g["prec"])
    p = _params_from_golden(g)
    with pytest.raises(api.J2kHipError, match="no encode in progress"):
        enc.encode_end()
    enc.encode_begin_host(frame, lay, p)
    with pytest.raises(api.J2kHipError, match="not been finished"):
        enc.encode_begin_host(frame, lay, p)
    # the failed second _begin has drained the handle: it is free again and still exact
    assert enc.encode_host(frame, lay, p) == cs


def _ae16_planes(w, h, nc, seed):
    """After Effects "15+1-bit" samples: 0..32768 inclusive (reference: src/aftereffects/FrameSeq.cpp:311-314)."""
    pl = synth.planes(w, h, nc, 15, seed, "A")
    pl[:, ::7, ::5] = 32768  # white
    pl[:, 1::9, 2::11] = 16384
    pl[:, 2::13, 1::3] = 16385
    pl[:, 3::17, ::19] = 0
    return pl


PROMOTE = [(300, 200, 3, True, True, 0), (300, 200, 3, False, True, 0), (150, 130, 4, True, True, 64), (257, 131, 4, False, True, 0),
           (640, 260, 4, False, True, 0), (130, 70, 1, True, False, 0), (300, 200, 3, False, False, 128)]


@pytest.mark.parametrize("case", PROMOTE, ids=str)
def test_promote_ae16_whole_codestream(enc, oracle, case):
    """promote_ae16 = 1 through the whole path: an ARGB64 world of 15+1-bit samples encodes to the bytes the
    reference's PromoteWorld + WriteFile would produce, i.e. oracle.encode(Promote(samples)) -- RGB and RGBA, 5/3 and
    9/7, tiled and not, from host and from device frames."""
    api = _api()
    from oracle.oracle import make_params
    w, h, nc, rev, mct, tile = case
    pl = _ae16_planes(w, h, nc, 808 + w)
    v = pl.astype(np.int64)
    promoted = np.where(v > 16384, ((v - 1) << 1) + 1, v << 1).astype(np.int32)
    assert promoted.max() == 65535 and promoted.min() == 0
    ref = oracle.encode(promoted, make_params(w, h, nc, 16, reversible=rev, mct=mct and nc >= 3, numres=4, tile=tile))
    frame, lay = synth.ae_frame(pl, 16, row_pad_bytes=16)
    p = api.make_params(w, h, nc, 16, reversible=rev, ycc=mct and nc >= 3, num_resolutions=4, tile_size=tile, promote=True)
    assert enc.encode_host(frame, lay, p) == ref
    d = enc.upload(frame)
    try:
        assert enc.encode_device(d, lay, p)[2] == ref
        api.tune("no_fuse", 1)  # and through the stand-alone front end
        try:
            assert enc.encode_device(d, lay, p)[2] == ref
        finally:
            api.tune("no_fuse", 0)
    finally:
        enc.free(d)
    if tile == 0:
        # and through the plug-in's Codec interface: HipCodec(.., PromoteAE16) lets the AE layer drop PromoteWorld / DemoteWorld
        from oracle.oracle import strip_com
        os.environ["J2K_HOST_TEST_PROMOTE"] = "1"
        try:
            got, err = _host_write(frame, lay, w, h, nc, 16, reversible=rev, ycc=mct and nc >= 3, layers=1, tile=0, honour=True)
        finally:
            del os.environ["J2K_HOST_TEST_PROMOTE"]
        assert got is not None, err
        full = oracle.encode(promoted, make_params(w, h, nc, 16, reversible=rev, mct=mct and nc >= 3, numres=6))
        assert strip_com(got) == full


def test_sequence_of_c5_frames_is_frame_by_frame(golden):
    """BASELINE config 5 the way bench.py --mode c5 drives it: 8 frames of 4096 x 2160 RGB10 in one
    j2k_hip_encode_sequence_device call.  Frame 0 is libopenjp2's codestream (hash), every other frame is the
    codestream it gets on its own."""
    api = _api()
    name = "c5_frame0_4096x2160_rgb10_97"
    g = golden[name]
    W, H, prec = g["width"], g["height"], g["prec"]
    p = api.make_params(W, H, 3, prec, reversible=False, ycc=True, comment="")
    e = api.Encoder(0)
    dptrs, singles, lay = [], [], None
    try:
        for f in range(8):
            fr, lay = synth.ae_frame(synth.planes(W, H, 3, prec, 45678 + f), prec)
            d = e.upload(fr)
            dptrs.append(d)
            dptr, n, _ = e.encode_device(d, lay, p, download=False)
            singles.append((n, hashlib.sha256(e.d2h(dptr, n)).hexdigest()))
        assert singles[0] == (g["length"], g["sha256"])
        assert len({s[1] for s in singles}) == 8
        seq = e.encode_sequence_device(dptrs, lay, p, download=False)
        got = [(n, hashlib.sha256(e.d2h(dptr, n)).hexdigest()) for dptr, n, _ in seq]
        assert got == singles
    finally:
        for d in dptrs:
            e.free(d)
        e.close()


def test_c4_middle_tile_rows_through_the_base_pointer_offset(golden):
    """BASELINE config 4 as ranks 1..N-1 of the tile-sharded job see it (bench.py --mode c4): only the rank's own
    rows are resident, the channel views are offset backwards so that they describe the whole image, and the tiles
    encoded sit at a non-zero y-origin.  Tile rows 1..2 (tiles 8..23) of a 16384 x 6144 image, 2048^2 tiles, against the
    tile-part bytes libopenjp2 wrote for them; then tile row 0 the same way."""
    api = _api()
    name = "c4_rows_16384x6144_rgb16_53_tile2048"
    if name not in golden:
        pytest.skip("full-size golden not generated")
    g = golden[name]
    W, H, T, prec = g["width"], g["height"], 2048, g["prec"]
    pl = synth.planes(W, H, 3, prec, g["seed"], g["dist"])
    p = api.make_params(W, H, 3, prec, reversible=True, ycc=True, tile_size=T, comment="")
    e = api.Encoder(0)
    try:
        for first, count in ((8, 16), (0, 8)):
            r0, r1 = first // 8 * T, (first + count) // 8 * T
            frame, lay = synth.ae_frame(pl[:, r0:r1], prec)
            d = e.upload(frame)
            del frame
            try:
                base = d - r0 * lay["rowbytes"]  # rows above the rank's share do not exist
                tp = e.encode_tiles_device(base, lay, p, first, count)
            finally:
                e.free(d)
            exp = g["tileparts"][f"{first}:{count}"]
            assert len(tp) == exp["length"]
            assert hashlib.sha256(tp).hexdigest() == exp["sha256"]
    finally:
        e.close()


def test_stage_t1_refuses_overlapping_rectangles(enc):
    """The modeller rewrites each block of the coefficient plane in place: overlapping rectangles are an error, not silent garbage."""
    api = _api()
    coef = np.zeros((64, 128), dtype=np.int32)
    with pytest.raises(api.J2kHipError, match="overlap"):
        enc.stage_t1(coef, [(0, 0, 64, 64), (32, 16, 64, 48)], [0, 0], [1.0, 1.0], True)
    with pytest.raises(api.J2kHipError, match="stride"):
        enc.stage_t1(coef, [(100, 0, 64, 64)], [0], [1.0], True)
    assert len(enc.stage_t1(coef, [(0, 0, 64, 64), (64, 0, 64, 64)], [0, 1], [1.0, 1.0], True)) == 2

"""The decode path: j2k_hip_read_info / j2k_hip_decode and HipCodec::GetFileInfo / ::ReadFile against
the decode oracle (pinned to libopenjp2's decoded planes, tests/test_decode_oracle.py) and the committed hashes.
Reference path: OpenJPEGCodec::GetFileInfo / ::ReadFile, src/common/j2k_openjpeg_codec.cpp:222-426, :451-586."""
import ctypes as C
import glob
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, golden_case
from j2k_amd import api, synth

FILES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN_DIR, "*.j2k")) + glob.glob(os.path.join(GOLDEN_DIR, "*.jp2")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load(fname):
    return open(os.path.join(GOLDEN_DIR, fname), "rb").read()


# ------------------------------------------------------------------------------------------------ CPU: headers
@pytest.mark.parametrize("fname", FILES)
def test_read_info_matches_oracle(oracle, golden, fname):
    data = load(fname)
    i, o = api.read_info(data), oracle.decode_info(data)
    assert (i["width"], i["height"], i["channels"], i["depth"]) == (o["width"], o["height"], o["ncomp"], o["prec"])
    assert (i["reversible"], i["ycc"], i["num_resolutions"]) == (o["reversible"], o["mct"], o["numres"])
    assert i["file_format"] == o["jp2"]
    assert (i["icc_profile_offset"], i["icc_profile_len"]) == ((o["icc_off"], o["icc_len"]) if o["icc_len"] else (0, 0))
    assert i["alpha"] == (o["alpha_mask"].bit_length() if o["alpha_mask"] else 0)
    g = golden[fname.rsplit(".", 1)[0]]
    if fname.endswith(".jp2") and not g.get("icc_len"):
        assert i["color_space"] == g["color_space"]


def test_read_info_rejects_garbage():
    for junk in (b"", b"\x00" * 40, b"\xff\x4f\xff\x51\x00\x10", b"\x00\x00\x00\x0cjP  \r\n\x87\n" + b"\x00" * 30):
        with pytest.raises(api.J2kHipError):
            api.read_info(junk)
    data = bytearray(load("g1_64x64_grey_1lvl.j2k"))
    data[data.index(b"\xff\x52") + 12] = 0x44  # code-block style bits that do not exist
    with pytest.raises(api.J2kHipError, match="code-block style"):
        api.read_info(bytes(data))
    data[data.index(b"\xff\x52") + 12] = 0x04  # termination on each pass: a style like any other since round 3
    assert api.read_info(bytes(data))["width"] == 64


# ------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def enc():
    e = api.Encoder(0)
    yield e
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fname", FILES)
def test_decode_equals_libopenjp2(enc, oracle, golden, fname):
    """Every committed file, at full size and at every reduced size libopenjp2 was asked for: the decoded samples are
    libopenjp2's (hash), hence the oracle's; the lossless ones are the generator's image."""
    name = fname.rsplit(".", 1)[0]
    g = golden[name]
    data = load(fname)
    dec = enc.decode_planar(data)
    assert dec.shape == (g["ncomp"], g["height"], g["width"])
    assert sha(dec.astype(np.int32)) == g["decoded_sha256"], name
    if g["params"].get("reversible", True) and "rates" not in g and "psnr_targets" not in g:
        assert np.array_equal(dec, synth.planes(g["width"], g["height"], g["ncomp"], g["prec"], g["seed"], g["dist"]))
    for r, h in g["decoded_reduced_sha256"].items():
        red = enc.decode_planar(data, subsample=1 << int(r))
        assert red.shape == (g["ncomp"], -(-g["height"] >> int(r)), -(-g["width"] >> int(r)))
        assert sha(red.astype(np.int32)) == h, (name, r)


@pytest.mark.gpu
def test_decode_random_shapes_against_oracle(enc, oracle):
    """Encode on the GPU, decode on the GPU, compare with the oracle's decode of the same bytes: odd sizes, tiles with odd
    origins, 1..4 components, every precision class, both wavelets, several layers cut by the rate allocation."""
    rng = np.random.default_rng(int(os.environ.get("J2K_FUZZ_SEED", "77")))  # (more cases, other seeds, code-block sizes: by hand)
    for i in range(int(os.environ.get("J2K_FUZZ_CASES", "12"))):
        w, h = int(rng.integers(20, 300)), int(rng.integers(20, 260))
        nc = int(rng.choice([1, 3, 4]))
        prec = int(rng.choice([8, 10, 12, 16]))
        rev = bool(rng.integers(0, 2))
        numres = int(rng.integers(1, 6))
        tile = int(rng.choice([0, 0, 64, 100]))
        if tile and tile < (1 << (numres - 1)):
            tile = 0
        rates = [float(x) for x in ([40, 10], [25], None, None)[int(rng.integers(0, 4))] or []] or None
        pl = synth.planes(w, h, nc, prec, 500 + i, "AB"[i & 1])
        frame, lay = synth.ae_frame(pl, prec)
        cb = (64, 64) if "J2K_FUZZ_SEED" not in os.environ else (int(rng.choice([16, 32, 64])), int(rng.choice([16, 32, 64])))
        p = api.make_params(w, h, nc, prec, reversible=rev, ycc=nc >= 3, num_resolutions=numres, tile_size=tile, rates=rates,
                            progression=int(rng.integers(0, 5)) if not tile else 0, cblk=cb)
        cs = enc.encode_host(frame, lay, p)
        for sub in (1, 2):
            if sub > 1 and numres < 2:
                continue
            ref = oracle.decode(cs, sub.bit_length() - 1)
            got = enc.decode_planar(cs, subsample=sub)
            assert np.array_equal(got.astype(np.int32), ref), (w, h, nc, prec, rev, numres, tile, rates, sub)
        if rev and not rates:
            assert np.array_equal(enc.decode_planar(cs), pl)


@pytest.mark.gpu
@pytest.mark.parametrize("device", [False, True], ids=["host", "device"])
def test_decode_into_ae_frame_touches_only_the_channels(enc, golden, device):
    """The destination is the host's interleaved ARGB frame (WorldToBuffer layout, row padding): three channels are
    decoded, the A samples and the row padding keep their bytes -- CopyBuffer writes samples, nothing else."""
    for name in ("g3_300x200_rgb8_53_rct", "g4_300x200_rgb16_53_rct_tile128"):
        g, pl, _, cs = golden_case(golden, name)
        w, h, prec = g["width"], g["height"], g["prec"]
        ref, lay = synth.ae_frame(pl, prec, row_pad_bytes=12)
        frame = np.full_like(ref, 0xA5)
        enc.decode_ae(cs, frame, lay, w, h, 3, device=device)
        sb, rb = lay["sample_bytes"], lay["rowbytes"]
        px = np.lib.stride_tricks.as_strided(frame, shape=(h, w, 4 * sb), strides=(rb, 4 * sb, 1))
        rx = np.lib.stride_tricks.as_strided(ref, shape=(h, w, 4 * sb), strides=(rb, 4 * sb, 1))
        assert np.array_equal(px[:, :, sb:], rx[:, :, sb:])              # R, G, B samples
        assert (px[:, :, :sb] == 0xA5).all()                              # A untouched
        pad = np.lib.stride_tricks.as_strided(frame[4 * sb * w:], shape=(h, 12), strides=(rb, 1))
        assert (pad == 0xA5).all()
    # four channels without row padding: every byte is a decoded sample (the direct path)
    g, pl, _, cs = golden_case(golden, "g9_300x200_rgba8_53_rct")
    ref, lay = synth.ae_frame(pl, 8)
    frame = np.zeros_like(ref)
    enc.decode_ae(cs, frame, lay, 300, 200, 4, device=device)
    assert np.array_equal(frame, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("name,bits,depth", [("g6_300x200_rgb16_97_ict", 8, 8), ("g7_300x200_rgb10_53", 16, 16), ("g7_300x200_rgb10_53", 8, 8),
                                             ("g3_300x200_rgb8_53_rct", 16, 16), ("g3_300x200_rgb8_53_rct", 16, 12), ("g9_97x61_grey12_97_4lvl", 16, 16)])
def test_decode_depth_conversion_is_copychannel(enc, oracle, golden, name, bits, depth):
    """Destination depth != the file's precision: CopyChannel's down-shift / bit-replicating up-shift
    (src/common/j2k_codec.cpp:254-371), here inside the output kernel."""
    g, _, _, cs = golden_case(golden, name)
    got = enc.decode_planar(cs, sample_bits=bits, depth=depth)
    dec = oracle.decode(cs)
    w, h = g["width"], g["height"]
    sb = bits // 8
    for c in range(g["ncomp"]):
        exp = oracle.copy_channel_out(dec[c], g["prec"], sb, depth, sb, w * sb, w, h)
        exp = exp.view(np.uint16 if sb == 2 else np.uint8).reshape(h, w)
        assert np.array_equal(got[c], exp), (name, c)


@pytest.mark.gpu
def test_decode_into_scattered_destinations(enc, golden):
    """The C ABI's destination is one strided view per channel: separately allocated planar buffers (spans of their
    own, copied straight), a planar channel with padded rows, bottom-up rows, and the samples of 3-byte pixels -- every
    byte that is not a channel sample keeps its contents."""
    g, pl, _, cs = golden_case(golden, "g3_300x200_rgb8_53_rct")
    w, h = g["width"], g["height"]
    # (1) three buffers of their own, far from each other
    keep = [np.zeros(1 << 20, dtype=np.uint8) for _ in range(2)]  # (whatever the allocator puts between them)
    a, b, c = np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint8)
    enc.decode_channels(cs, [a, b, c])
    assert np.array_equal(np.stack([a, b, c]), pl)
    del keep
    # (2) padded rows, bottom-up rows, and a channel inside 3-byte pixels, all in one call
    pad = np.full((h, w + 13), 0x5A, np.uint8)
    flip = np.full((h, w), 0x5A, np.uint8)
    pix = np.full((h, w, 3), 0x5A, np.uint8)
    enc.decode_channels(cs, [pad[:, :w], flip[::-1], pix[:, :, 1]])
    assert np.array_equal(pad[:, :w], pl[0]) and (pad[:, w:] == 0x5A).all()
    assert np.array_equal(flip[::-1], pl[1])
    assert np.array_equal(pix[:, :, 1], pl[2]) and (pix[:, :, 0] == 0x5A).all() and (pix[:, :, 2] == 0x5A).all()
    # (3) the three samples of 3-byte pixels: interleaved, every byte a sample
    pix = np.zeros((h, w, 3), np.uint8)
    enc.decode_channels(cs, [pix[:, :, 0], pix[:, :, 1], pix[:, :, 2]])
    assert np.array_equal(pix.transpose(2, 0, 1), pl)
    # (4) a 16-bit file into reused planar buffers
    g, pl, _, cs = golden_case(golden, "g4_300x200_rgb16_53_rct_tile128")
    out = np.full((3, 200, 300), 0xBEEF, np.uint16)
    assert enc.decode_planar(cs, out=out) is out and np.array_equal(out, pl)


@pytest.mark.gpu
def test_decode_big_frame_into_ae_world_in_bands(enc):
    """A frame big enough for the banded download (the host merges band k into the ARGB64 world while band k + 1 is on its
    way): R, G, B land where they belong, A and the row padding keep their bytes."""
    w, h, prec = 3300, 2700, 16  # 71 MB of pixels: two bands
    pl = synth.planes(w, h, 3, prec, 21)
    src, lay = synth.ae_frame(pl, prec, row_pad_bytes=24)
    cs = enc.encode_host(src, lay, api.make_params(w, h, 3, prec, reversible=True, ycc=True, comment=""))
    frame = np.full_like(src, 0xA5)
    enc.decode_ae(cs, frame, lay, w, h, 3)
    rb = lay["rowbytes"]
    px = np.lib.stride_tricks.as_strided(frame, shape=(h, w, 8), strides=(rb, 8, 1))
    rx = np.lib.stride_tricks.as_strided(src, shape=(h, w, 8), strides=(rb, 8, 1))
    assert np.array_equal(px[:, :, 2:], rx[:, :, 2:]) and (px[:, :, :2] == 0xA5).all()
    assert (np.lib.stride_tricks.as_strided(frame[8 * w:], shape=(h, 24), strides=(rb, 1)) == 0xA5).all()


@pytest.mark.gpu
def test_decode_smaller_destination_and_fewer_channels(enc, oracle, golden):
    """"CopyBuffer is based on the destination size" (j2k_openjpeg_codec.cpp:496-499): a destination smaller than the
    image receives its top-left part; fewer destination channels than components receive the first ones."""
    g, _, _, cs = golden_case(golden, "g9_300x200_rgba8_53_rct")
    dec = oracle.decode(cs)
    out = np.zeros((2, 120, 200), dtype=np.uint8)
    arr = (api.OutPlane * 2)()
    for c in range(2):
        arr[c].base = out.ctypes.data + c * 120 * 200
        arr[c].colbytes, arr[c].rowbytes, arr[c].sample_bits, arr[c].depth, arr[c].width, arr[c].height = 1, 200, 8, 8, 200, 120
    buf = np.frombuffer(cs, dtype=np.uint8)
    enc._check(enc.L.j2k_hip_decode(enc.h, buf.ctypes.data, len(cs), 1, arr, 2))
    assert np.array_equal(out, dec[:2, :120, :200])


@pytest.mark.gpu
def test_decode_errors(enc):
    cs = load("g1_64x64_grey_1lvl.j2k")
    with pytest.raises(api.J2kHipError, match="resolutions"):
        enc.decode_planar(cs, subsample=4)
    with pytest.raises(api.J2kHipError):
        enc.decode_planar(cs[:40])
    # a file cut short decodes what is there (like libopenjp2's default, non-strict mode)
    full = load("g3_300x200_rgb8_53_rct.j2k")
    assert enc.decode_planar(full[:len(full) // 2]).shape == (3, 200, 300)
    # and the handle still works
    assert enc.decode_planar(cs).shape == (1, 64, 64)


def _host():
    api.load_library()
    H = C.CDLL(os.path.join(os.path.dirname(api.LIBPATH), "libj2k_host.so"))
    H.j2k_host_test_read.restype = C.c_long
    H.j2k_host_test_read.argtypes = [C.c_void_p, C.c_ulong, C.c_uint, C.c_void_p, C.c_uint, C.c_uint, C.c_long, C.c_int, C.c_int, C.c_int,
                                     C.c_char_p, C.c_ulong]
    H.j2k_host_test_info.restype = C.c_long
    H.j2k_host_test_info.argtypes = [C.c_void_p, C.c_ulong, C.POINTER(C.c_long), C.c_void_p, C.c_ulong, C.c_char_p, C.c_ulong]
    return H


def test_hip_codec_get_file_info(golden):
    """HipCodec::GetFileInfo through the Codec interface (no device needed): the FileInfo fields the reference fills
    (j2k_openjpeg_codec.cpp:292-377), incl. its malloc'd copy of the ICC profile and the cdef-derived channel map."""
    H = _host()
    for fname in FILES:
        g = golden[fname.rsplit(".", 1)[0]]
        data = load(fname)
        buf = np.frombuffer(data, dtype=np.uint8)
        out = (C.c_long * 13)()
        icc = np.zeros(4096, dtype=np.uint8)
        err = C.create_string_buffer(256)
        assert H.j2k_host_test_info(buf.ctypes.data, len(data), out, icc.ctypes.data, 4096, err, 256) == 0, err.value
        assert (out[0], out[1], out[2], out[3]) == (g["width"], g["height"], g["ncomp"], g["prec"])
        assert out[4] == (2 if fname.endswith(".jp2") else 1)  # j2k::JP2 / j2k::J2C
        assert bool(out[8]) == g["params"].get("reversible", True)
        if g.get("icc_len"):
            assert out[7] == g["icc_len"] and out[5] == (10 if g["ncomp"] >= 3 else 9)  # iccRGB / iccLUM
            x, exp = g["icc_seed"], bytearray()
            for _ in range(g["icc_len"]):
                x = (x * 1103515245 + 12345) & 0x7fffffff
                exp.append((x >> 16) & 0xff)
            assert icc[:out[7]].tobytes() == bytes(exp)
        elif fname.endswith(".jp2"):
            assert out[5] == {1: 1, 2: 2, 3: 3, 4: 5, 5: 7, 0: 0}[g["color_space"]]  # sRGB, sLUM, sYCC, esYCC, CMYK
            if g.get("alpha_channel", -1) >= 0:
                assert out[6] == 3 and out[9 + g["alpha_channel"]] == 3  # STRAIGHT, channelMap[..] = ALPHA
    err = C.create_string_buffer(256)
    junk = np.zeros(64, dtype=np.uint8)
    assert H.j2k_host_test_info(junk.ctypes.data, 64, (C.c_long * 13)(), None, 0, err, 256) == -1
    assert err.value.startswith(b"Can't read this format")


def test_hip_codec_reports_a_palette_in_file_info():
    """HipCodec::GetFileInfo on a palettised JP2: FileInfo.LUTsize / .LUT / .LUTmap as the reference fills them
    (j2k_openjpeg_codec.cpp:362-401: LUT[i].channel[c] = entry, LUTmap[i] = channelMap[cmap[i].pcol])."""
    from test_read_fallback import _with_palette
    H = _host()
    H.j2k_host_test_lut.restype = C.c_long
    H.j2k_host_test_lut.argtypes = [C.c_void_p, C.c_ulong, C.c_void_p, C.POINTER(C.c_long)]
    jp2 = load("j2_64x48_grey8.jp2")
    for column_of in ((0, 1, 2), (2, 0, 1)):
        crafted, pal = _with_palette(jp2, 200, 3, column_of)
        buf = np.frombuffer(crafted, dtype=np.uint8)
        lut = np.zeros((256, 4), dtype=np.uint8)
        lutmap = (C.c_long * 4)()
        assert H.j2k_host_test_lut(buf.ctypes.data, len(crafted), lut.ctypes.data, lutmap) == 200
        assert np.array_equal(lut[:200, :3], pal)
        assert tuple(lutmap[:3]) == column_of  # channelMap = RED, GREEN, BLUE = ChannelName 0, 1, 2
    plain = np.frombuffer(jp2, dtype=np.uint8)
    assert H.j2k_host_test_lut(plain.ctypes.data, len(jp2), np.zeros((256, 4), dtype=np.uint8).ctypes.data, (C.c_long * 4)()) == 0


@pytest.mark.gpu
def test_hip_codec_read_file(golden, oracle):
    """HipCodec::ReadFile driven like the plug-in drives a Codec: ARGB destination of the subsampled size, error
    convention (j2k::Exception("Error reading file"))."""
    H = _host()
    for name, sub in (("g3_300x200_rgb8_53_rct", 1), ("g6_300x200_rgb16_97_ict", 1), ("g6_300x200_rgb16_97_ict", 2),
                      ("g9_300x200_rgba8_53_rct", 4), ("g4_300x200_rgb16_53_rct_tile128", 2)):
        g, pl, _, cs = golden_case(golden, name)
        red = sub.bit_length() - 1
        dec = oracle.decode(cs, red)
        nc, h, w = dec.shape
        ref, lay = synth.ae_frame(dec, g["prec"])
        sb = lay["sample_bytes"]
        frame = np.zeros_like(ref)
        if nc < 4:
            view = np.lib.stride_tricks.as_strided(frame, shape=(h, w, sb), strides=(lay["rowbytes"], 4 * sb, 1))
            view[:] = 0xff  # the generator's frame carries opaque alpha there
        buf = np.frombuffer(cs, dtype=np.uint8)
        err = C.create_string_buffer(256)
        # destination depth = the container's depth (AE hands 8- or 16-bit worlds): the decoded precision is scaled up to it
        rc = H.j2k_host_test_read(buf.ctypes.data, len(cs), sub, frame.ctypes.data, w, h, lay["rowbytes"], sb, nc, 8 * sb, err, 256)
        assert rc == 0, err.value
        exp = np.zeros_like(ref)
        for c in range(nc):
            off = lay["channel_offsets"][(1, 2, 3, 0)[c]]
            ch = oracle.copy_channel_out(dec[c], g["prec"], sb, 8 * sb, 4 * sb, lay["rowbytes"], w, h)
            exp[off:] |= ch[:len(exp) - off] if off else ch
        if nc < 4:
            ev = np.lib.stride_tricks.as_strided(exp, shape=(h, w, sb), strides=(lay["rowbytes"], 4 * sb, 1))
            ev[:] = 0xff
        assert np.array_equal(frame, exp), (name, sub)
    err = C.create_string_buffer(256)
    cs = load("g1_64x64_grey_1lvl.j2k")
    buf = np.frombuffer(cs, dtype=np.uint8)
    frame = np.zeros(64 * 64 * 4, dtype=np.uint8)
    assert H.j2k_host_test_read(buf.ctypes.data, len(cs), 8, frame.ctypes.data, 8, 8, 32, 1, 1, 8, err, 256) == -1
    assert err.value.startswith(b"Error reading file") and b"resolutions" in err.value


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c2_4096_rgb8_97", "c4_tile_2048_rgb16_53", "c5_frame0_4096x2160_rgb10_97", "c3_8192_rgb16_97_5lvl"])
def test_full_size_round_trip(enc, golden, name):
    """BASELINE-size frames: encode on the GPU (bytes = libopenjp2's, checked elsewhere), decode on the GPU, compare
    with the hash of libopenjp2's decode of the same codestream; the 5/3 one must give the input back."""
    if name not in golden:
        pytest.skip("full-size golden not generated")
    g, pl, _, _ = golden_case(golden, name)
    frame, lay = synth.ae_frame(pl, g["prec"])
    kw = g["params"]
    p = api.make_params(g["width"], g["height"], g["ncomp"], g["prec"], reversible=kw.get("reversible", True), ycc=kw.get("mct", False),
                        tile_size=kw.get("tile", 0), num_resolutions=kw.get("numres", 6), comment="")
    cs = enc.encode_host(frame, lay, p)
    assert hashlib.sha256(cs).hexdigest() == g["sha256"]
    dec = enc.decode_planar(cs)
    if kw.get("reversible", True):
        assert np.array_equal(dec, pl)
    assert sha(dec.astype(np.int32)) == g["decoded_sha256"]


@pytest.mark.gpu
def test_destroying_a_handle_that_decoded_returns_its_memory():
    """Every arena of a handle, the decode path's included, goes back to the device with j2k_hip_destroy."""
    import torch
    data = load("g6_300x200_rgb16_97_ict.j2k")

    def cycle():
        e = api.Encoder(0)
        try:
            e.decode_planar(data)
            pl = synth.planes(640, 480, 3, 8, 3)
            frame, lay = synth.ae_frame(pl, 8)
            e.encode_host(frame, lay, api.make_params(640, 480, 3, 8, reversible=False, ycc=True))
        finally:
            e.close()
    # what the runtime itself keeps is allocated after a few cycles: code objects, and one hardware queue per stream until its
    # pool of GPU_MAX_HW_QUEUES (24 here, tests/conftest.py) is full -- 2 MiB each, kept when the stream is destroyed
    for _ in range(14):
        cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for _ in range(12):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]
    assert free0 - free1 < (8 << 20), f"{(free0 - free1) >> 20} MiB lost over 12 create/decode/destroy cycles"


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [0, 2], ids=["wave-per-block", "lane-per-block"])
def test_both_tier1_decoders_agree_with_libopenjp2(enc, golden, lanes):
    """`t1dec_lanes`: 0 = a wave per code-block, 2 = a lane per code-block (64 blocks per wave), 1 (default) = chosen by
    the size of the file.  Both kernels on small files of every kind."""
    api.tune("t1dec_lanes", lanes)
    try:
        for fname in ("g6_300x200_rgb16_97_ict.j2k", "g4_300x200_rgb16_53_rct_tile128.j2k", "o3_200x300_rgba16_53_tile128_2layers_pcrl.j2k",
                      "q2_300x200_rgb8_97_ict_q30_38_45.j2k"):
            g = golden[fname.rsplit(".", 1)[0]]
            dec = enc.decode_planar(load(fname))
            assert sha(dec.astype(np.int32)) == g["decoded_sha256"], fname
            for r, h in g["decoded_reduced_sha256"].items():
                assert sha(enc.decode_planar(load(fname), subsample=1 << int(r)).astype(np.int32)) == h, (fname, r)
    finally:
        api.tune("t1dec_lanes", 1)


@pytest.mark.gpu
@pytest.mark.parametrize("tail", [1, 2, 3, 0], ids=["tail-by-size", "tail-half", "tail-third", "no-tail"])
def test_lane_per_block_decoder_on_every_golden_file(enc, golden, tail):
    """`t1dec_tail`: the heaviest blocks of a lane-per-block decode go to the wave-per-block kernel on a second stream
    (1 = chosen by a cost model, n >= 2 = the heaviest 1/n of the blocks: both kernels in every one of these decodes)."""
    api.tune("t1dec_lanes", 2)
    api.tune("t1dec_tail", tail)
    try:
        for fname in FILES:
            g = golden[fname.rsplit(".", 1)[0]]
            assert sha(enc.decode_planar(load(fname)).astype(np.int32)) == g["decoded_sha256"], fname
    finally:
        api.tune("t1dec_lanes", 1)
        api.tune("t1dec_tail", 1)

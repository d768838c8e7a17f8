"""JP2 file wrapper: oracle, C ABI and GPU path against whole files written by
libopenjp2's own JP2 writer (tests/golden/j*.jp2, made by tests/golden/make_golden.py).

The reference disables its JP2 branch (src/common/j2k_openjpeg_codec.cpp:609-614) because OpenJPEG
seeks back to patch the jp2c box; these tests also pin that our path delivers the file strictly
sequentially."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from j2k_amd import api, synth
from oracle.oracle import make_params

JP2 = ["j1_64x48_rgb8_srgb", "j2_64x48_grey8", "j3_64x48_rgba8_srgb_alpha", "j4_64x48_rgb16_icc",
       "j5_64x48_rgb8_sycc_97", "j6_40x30_greya8", "j7_40x30_cmyk8", "j8_40x30_rgb8_unspecified",
       "j9_40x30_rgba16_icc_alpha"]


def fake_icc(n, seed):  # same generator as tests/golden/make_golden.py
    x, out = seed, bytearray()
    for _ in range(n):
        x = (x * 1103515245 + 12345) & 0x7fffffff
        out.append((x >> 16) & 0xff)
    return bytes(out)


def case(golden, name):
    g = golden[name]
    pl = synth.planes(g["width"], g["height"], g["ncomp"], g["prec"], g["seed"], g["dist"])
    icc = fake_icc(g["icc_len"], g["icc_seed"]) if g["icc_len"] else None
    f = open(os.path.join(GOLDEN_DIR, name + ".jp2"), "rb").read()
    assert hashlib.sha256(f).hexdigest() == g["sha256"]
    return g, pl, icc, f


def hip_params(g, icc):
    kw = g["params"]
    return api.make_params(g["width"], g["height"], g["ncomp"], g["prec"], reversible=kw.get("reversible", True),
                           ycc=kw.get("mct", False), num_resolutions=kw.get("numres", 6), comment=g["comment"],
                           jp2=True, color_space=g["color_space"], alpha_channel=g["alpha_channel"], icc=icc)


def boxes(b):
    """[(type, payload)] of the top-level boxes."""
    out, off = [], 0
    while off < len(b):
        n = int.from_bytes(b[off:off + 4], "big")
        out.append((b[off + 4:off + 8], b[off + 8:off + n]))
        off += n
    return out


@pytest.mark.parametrize("name", JP2)
def test_oracle_jp2_matches_golden_file(oracle, golden, name):
    g, pl, icc, f = case(golden, name)
    p = make_params(g["width"], g["height"], g["ncomp"], g["prec"], **g["params"])
    cs = oracle.encode(pl, p, comment=g["comment"])
    assert oracle.jp2_wrap(cs, p, g["color_space"], icc, g["alpha_channel"]) == f


@pytest.mark.parametrize("name", JP2)
def test_cabi_file_header_matches_golden(golden, name):
    g, _, icc, f = case(golden, name)
    k = f.index(b"jp2c") + 4
    assert api.file_header(hip_params(g, icc), len(f) - k) == f[:k]


def test_file_header_raw_codestream_is_empty():
    assert api.file_header(api.make_params(64, 48, 3, 8), 1000) == b""


def test_file_header_structure_and_premultiplied_alpha():
    p = api.make_params(20, 10, 4, 16, num_resolutions=2, jp2=True, color_space=1, alpha_channel=3, alpha_premultiplied=True)
    h = api.file_header(p, 12345)
    bx = boxes(h + b"\0" * 12345)
    assert [t for t, _ in bx] == [b"jP  ", b"ftyp", b"jp2h", b"jp2c"]
    assert bx[0][1] == b"\r\n\x87\n" and bx[1][1] == b"jp2 \0\0\0\0jp2 "
    sub = boxes(bx[2][1])
    assert [t for t, _ in sub] == [b"ihdr", b"colr", b"cdef"]
    assert sub[0][1] == (10).to_bytes(4, "big") + (20).to_bytes(4, "big") + b"\x00\x04\x0f\x07\x00\x00"
    assert sub[1][1] == b"\x01\x00\x00\x00\x00\x00\x10"
    # channels 0..2 colour (Typ 0, Asoc 1..3); channel 3 premultiplied opacity (Typ 2) of the whole image (Asoc 0)
    assert sub[2][1] == bytes.fromhex("0004" "0000 0000 0001" "0001 0000 0002" "0002 0000 0003" "0003 0002 0000".replace(" ", ""))
    assert len(bx[3][1]) == 12345


def test_file_header_large_codestream_uses_xlbox():
    p = api.make_params(20, 10, 1, 8, num_resolutions=2, jp2=True, color_space=2)
    h = api.file_header(p, (1 << 32) + 5)
    assert h[-16:-8] == b"\x00\x00\x00\x01jp2c" and int.from_bytes(h[-8:], "big") == (1 << 32) + 5 + 16
    h = api.file_header(p, (1 << 32) - 9)
    assert h[-8:] == b"\xff\xff\xff\xffjp2c"


@pytest.mark.parametrize("kw", [dict(jp2=2), dict(jp2=True, color_space=6), dict(jp2=True, alpha_channel=3)])
def test_file_header_rejects_bad_wrapper_params(kw):
    p = api.make_params(20, 10, 3, 8, num_resolutions=2, **kw)
    with pytest.raises(api.J2kHipError):
        api.file_header(p, 10)


def test_icc_pointer_and_length_must_agree():
    p = api.make_params(20, 10, 3, 8, num_resolutions=2, jp2=True)
    p.icc_profile_len = 10
    with pytest.raises(api.J2kHipError):
        api.file_header(p, 10)


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", JP2)
def test_gpu_jp2_file_matches_golden(golden, name):
    g, pl, icc, f = case(golden, name)
    frame, lay = synth.ae_frame(pl, g["prec"]) if g["ncomp"] in (3, 4) else (None, None)
    enc = api.Encoder(0)
    p = hip_params(g, icc)
    if frame is not None:
        got = enc.encode_host(frame, lay, p)
        assert got == f
        assert enc.encode_host(frame, lay, p, via_sink=True) == f  # sequential sink writes only
    else:  # 1- and 2-channel images: planar host buffers
        got = enc.encode_planar_host(pl, p)
        assert got == f
    enc.close()


@pytest.mark.gpu
def test_gpu_jp2_decodes_with_libopenjp2(golden, opj):
    g, pl, icc, f = case(golden, "j3_64x48_rgba8_srgb_alpha")
    frame, lay = synth.ae_frame(pl, g["prec"])
    enc = api.Encoder(0)
    got = enc.encode_host(frame, lay, hip_params(g, icc))
    enc.close()
    dec, meta = opj.decode_ex(got)
    assert meta["jp2"] and meta["alpha_mask"] == 8 and meta["color_space"] == 1
    assert np.array_equal(dec, pl)


@pytest.mark.gpu
def test_gpu_jp2_device_entry_point_and_sharded_assembly(golden):
    """The framed device entry point emits the boxes itself; tile-sharded ranks emit tile-parts and rank 0
    prepends j2k_hip_file_header + main header."""
    from j2k_amd import sharding
    w, h = 160, 96
    pl = synth.planes(w, h, 3, 8, 5, "B")
    frame, lay = synth.ae_frame(pl, 8)
    enc = api.Encoder(0)
    d = enc.upload(frame)
    p = api.make_params(w, h, 3, 8, ycc=True, tile_size=64, num_resolutions=3, jp2=True, color_space=1)
    _, _, whole = enc.encode_device(d, lay, p)
    parts = [enc.encode_tiles_device(d, lay, p, t, 1) for t in range(6)]
    enc.close()
    assert sharding.assemble(p, parts) == whole
    assert whole[:12] == b"\x00\x00\x00\x0cjP  \r\n\x87\n"
    k = whole.index(b"jp2c")
    assert int.from_bytes(whole[k - 4:k], "big") == len(whole) - (k - 4)


@pytest.mark.gpu
def test_hip_codec_honour_settings_writes_jp2(golden):
    """HipCodec::WriteFile in HonourSettings mode: FileInfo.format/colorSpace/iccProfile/alpha select the wrapper
    (the reference's disabled JP2 branch); ReferenceLiteral keeps writing the raw codestream."""
    from oracle.oracle import strip_com
    api.load_library()
    H = C.CDLL(os.path.join(os.path.dirname(api.LIBPATH), "libj2k_host.so"))
    H.j2k_host_test_write_ex.restype = C.c_long
    H.j2k_host_test_write_ex.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_long] + [C.c_int] * 8 + [C.c_long, C.c_int, C.c_int,
                                         C.c_char_p, C.c_ulong, C.c_int, C.c_void_p, C.c_ulong, C.c_char_p, C.c_ulong]
    JP2_FMT, SRGB, ICC_RGB, STRAIGHT = 2, 1, 10, 3

    def write(name, honour, fmt, cspace, icc):
        g, pl, _, f = case(golden, name)
        frame, lay = synth.ae_frame(pl, g["prec"])
        out = np.empty(1 << 20, dtype=np.uint8)
        err = C.create_string_buffer(512)
        kw = g["params"]
        n = H.j2k_host_test_write_ex(frame.ctypes.data, g["width"], g["height"], lay["rowbytes"], lay["sample_bytes"],
                                     g["ncomp"], g["prec"], int(kw.get("reversible", True)), int(kw.get("mct", False)), 1, 0,
                                     int(honour), -1, fmt, cspace, icc, len(icc) if icc else 0, -1, out.ctypes.data,
                                     out.nbytes, err, 512)
        assert n >= 0, err.value
        return out[:n].tobytes(), f

    def split(f):  # (boxes before jp2c without the jp2c length, codestream)
        k = f.index(b"jp2c")
        return f[:k - 4], f[k + 4:]

    # default numresolution is 6 in the codec, the fixtures use fewer levels: compare the boxes, and the
    # codestream against the raw path with the same parameters
    for name, cspace, icc in [("j3_64x48_rgba8_srgb_alpha", SRGB, None),
                              ("j4_64x48_rgb16_icc", ICC_RGB, fake_icc(560, 104))]:
        got, f = write(name, True, JP2_FMT, cspace, icc)
        raw, _ = write(name, True, 1, cspace, icc)          # J2C: raw codestream
        lit, _ = write(name, False, JP2_FMT, cspace, icc)   # ReferenceLiteral ignores the format like the reference
        assert raw[:2] == b"\xff\x4f" and lit[:2] == b"\xff\x4f"
        gb, gcs = split(got)
        fb, _ = split(f)
        assert gb == fb
        assert gcs == raw
        k = got.index(b"jp2c")
        assert int.from_bytes(got[k - 4:k], "big") == 8 + len(raw)


def _boxes(data, start=0, end=None):
    """[(type, payload offset, payload length)] of the boxes in data[start:end]."""
    out, pos, end = [], start, len(data) if end is None else end
    while pos + 8 <= end:
        n, t = int.from_bytes(data[pos:pos + 4], "big"), data[pos + 4:pos + 8]
        if n == 0:
            n = end - pos
        out.append((t, pos + 8, n - 8))
        pos += n
    return out


def test_resolution_box_structure(oracle):
    """FileInfo.pixelAspect / .dpi (src/common/j2k_codec.h:168-169): a `res ` super-box with a capture-resolution box at
    the end of the JP2 header when the pixels are not square or a dpi is given; none otherwise (the file OpenJPEG writes).
    OpenJPEG has no writer for this box, so the check is structural (T.800 I.5.3.7) -- and the file must still read."""
    plain = api.file_header(api.make_params(64, 48, 3, 8, jp2=True, color_space=1), 1000)
    assert api.file_header(api.make_params(64, 48, 3, 8, jp2=True, color_space=1, pixel_aspect=(1, 1)), 1000) == plain
    for aspect, dpi in (((10, 11), 0.0), (None, 300.0), ((2, 1), 150.0)):
        h = api.file_header(api.make_params(64, 48, 3, 8, jp2=True, color_space=1, pixel_aspect=aspect, dpi=dpi), 1000)
        top = dict((t, (o, n)) for t, o, n in _boxes(h[:len(h) - 8]))
        inner = _boxes(h, top[b"jp2h"][0], top[b"jp2h"][0] + top[b"jp2h"][1])
        assert [t for t, _, _ in inner] == [b"ihdr", b"colr", b"res "]
        _, ro, rn = inner[-1]
        (t, o, n), = _boxes(h, ro, ro + rn)
        assert t == b"resc" and n == 10
        vn, vd, hn, hd = (int.from_bytes(h[o + 2 * i:o + 2 * i + 2], "big") for i in range(4))
        ve, he = (int.from_bytes(h[o + 8 + i:o + 9 + i], "big", signed=True) for i in range(2))
        v, hz = vn / vd * 10.0 ** ve, hn / hd * 10.0 ** he
        assert abs(v - (dpi or 72.0) / 0.0254) / v < 1e-3
        if aspect:
            assert abs(v / hz - aspect[0] / aspect[1]) < 1e-3  # a pixel `aspect` times as wide as high: fewer columns per metre
        else:
            assert abs(v - hz) < 1e-9
        assert h[len(plain) - 8 + (len(h) - len(plain)):] == plain[-8:]  # the jp2c header follows unchanged
    with pytest.raises(api.J2kHipError):
        api.file_header(api.make_params(64, 48, 3, 8, jp2=True, pixel_aspect=(3, 0)), 10)

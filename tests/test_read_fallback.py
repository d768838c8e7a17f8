"""The read side as the plug-in's DEFAULT codec (missing 1).  "HIP" sorts before "OpenJPEG" in the registry
(reference: src/common/j2k_codec.cpp:518, :540-548), so RGBAinputFile (src/common/j2k_rgba_file.cpp:41) hands every file
to HipCodec.  Files that use a JPEG 2000 feature the GPU decoder does not implement must reach the fallback reader
(HipCodec::SetFallback -- the plug-in passes its OpenJPEGCodec) for GetFileInfo and ReadFile alike; malformed files must
not; without a fallback they fail with the reference's "Error reading file".  Files: tests/golden/ext/ -- written by
libopenjp2's general encoder (tests/golden/make_golden.py --ext), features the plug-in's own writer never produces."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from j2k_amd import api

EXT = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "ext", "*.j2k")))
J2K_HIP_ERR_PARAM, J2K_HIP_ERR_UNSUPPORTED = 1, 6


def _host():
    api.load_library()
    H = C.CDLL(os.path.join(os.path.dirname(api.LIBPATH), "libj2k_host.so"))
    H.j2k_host_test_read_fallback.restype = C.c_long
    H.j2k_host_test_read_fallback.argtypes = [C.c_void_p, C.c_ulong, C.c_int, C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.POINTER(C.c_long),
                                              C.c_char_p, C.c_ulong]
    return H


def _read(H, data, with_fallback, w, h, nc):
    buf = np.frombuffer(data, dtype=np.uint8)
    frame = np.zeros(nc * w * h, dtype=np.uint8)
    info = (C.c_long * 3)()
    err = C.create_string_buffer(512)
    rc = H.j2k_host_test_read_fallback(buf.ctypes.data, len(data), int(with_fallback), frame.ctypes.data, w, h, nc, info, err, 512)
    return rc, list(info), frame, err.value.decode()


def _ext(name):
    return open(os.path.join(GOLDEN_DIR, "ext", name + ".j2k"), "rb").read()


def test_ext_fixtures_are_present():
    assert len(EXT) >= 9


@pytest.mark.parametrize("name", EXT)
def test_every_ext_file_is_decoded_or_reaches_the_fallback(golden, name):
    """No file libopenjp2 wrote may end as "Error reading file" when the plug-in has installed its other reader: either the
    header parses as supported (then the GPU tests hold the decode to libopenjp2's samples) or the status is UNSUPPORTED
    and both GetFileInfo and ReadFile arrive at the fallback.  No device is needed to tell."""
    data = _ext(name)
    g = golden[name]
    try:
        info = api.read_info(data)
    except api.J2kHipError as e:
        assert e.code == J2K_HIP_ERR_UNSUPPORTED, (name, str(e))
        H = _host()
        rc, inf, frame, err = _read(H, data, True, g["width"], g["height"], g["ncomp"])
        assert rc == 1, (name, rc, err)
        assert inf == [4242, 2424, 3]                      # the fallback's GetFileInfo answered
        assert frame[0] == 0xC0                            # and its ReadFile wrote the destination
        rc, _, _, err = _read(H, data, False, g["width"], g["height"], g["ncomp"])
        assert rc == -1 and err.startswith("Error reading file"), (name, err)
    else:
        assert (info["width"], info["height"], info["channels"]) == (g["width"], g["height"], min(g["ncomp"], 4))


def test_malformed_files_never_reach_the_fallback():
    H = _host()
    good = open(os.path.join(GOLDEN_DIR, "g3_300x200_rgb8_53_rct.j2k"), "rb").read()
    for data in (good[:60], good[:2] + b"\xff\x51\x00\x03" + good[6:], b"\xff\x4f\xff\x51" + bytes(40)):
        rc, _, frame, err = _read(H, data, True, 300, 200, 3)
        assert rc == -1 and err.startswith("Error reading file"), err
        assert frame[0] == 0
    with pytest.raises(api.J2kHipError) as ei:
        api.read_info(good[:60])
    assert ei.value.code == J2K_HIP_ERR_PARAM


def test_a_feature_patched_into_a_supported_file_is_unsupported():
    """More than sixteen components: UNSUPPORTED, not a parse error (up to sixteen the first four are read).  An RGN marker segment (a region of interest by
    MAXSHIFT) is read since round 3; a shift that no 30 bit-planes can hold is malformed."""
    good = bytearray(open(os.path.join(GOLDEN_DIR, "g3_300x200_rgb8_53_rct.j2k"), "rb").read())
    i = good.index(b"\xff\x5c")
    assert api.read_info(bytes(good[:i]) + b"\xff\x5e\x00\x05\x00\x00\x03" + bytes(good[i:]))["width"] == 300   # RGN, component 0, shift 3
    with pytest.raises(api.J2kHipError) as ei:
        api.read_info(bytes(good[:i]) + b"\xff\x5e\x00\x05\x00\x00\x40" + bytes(good[i:]))                      # shift 64
    assert ei.value.code == J2K_HIP_ERR_PARAM
    siz = good.index(b"\xff\x51")
    L = int.from_bytes(good[siz + 2:siz + 4], "big")

    def with_comps(n):  # SIZ rewritten to n components (the file's packets stay those of three)
        return bytes(good[:siz + 2]) + (L + 3 * (n - 3)).to_bytes(2, "big") + bytes(good[siz + 4:siz + 2 + L - 9 - 2]) + n.to_bytes(2, "big") + \
            bytes(good[siz + 2 + L - 9:siz + 2 + L]) + bytes([7, 1, 1] * (n - 3)) + bytes(good[siz + 2 + L:])
    assert api.read_info(with_comps(5))["channels"] == 4     # the first four are read, like the reference (src/common/j2k_openjpeg.cpp:278)
    assert api.read_info(with_comps(16))["channels"] == 4
    with pytest.raises(api.J2kHipError) as ei:
        api.read_info(with_comps(17))
    assert ei.value.code == J2K_HIP_ERR_UNSUPPORTED and "components" in str(ei.value)


@pytest.mark.gpu
def test_supported_files_stay_on_the_gpu_with_a_fallback_installed(golden, oracle):
    H = _host()
    name = "g3_300x200_rgb8_53_rct"
    data = open(os.path.join(GOLDEN_DIR, name + ".j2k"), "rb").read()
    rc, inf, frame, err = _read(H, data, True, 300, 200, 3)
    assert rc == 0, err
    assert inf == [300, 200, 3]
    assert np.array_equal(frame.reshape(3, 200, 300), oracle.decode(data).astype(np.uint8))


@pytest.mark.gpu
def test_a_six_component_file_reads_its_first_four_through_the_host_call(golden):
    """GetFileInfo says four channels and ReadFile fills them from components 0..3 (src/common/j2k_openjpeg.cpp:278, :530);
    the file is reversible, so those are the generator's planes."""
    from j2k_amd import synth
    name = "m1_90x70_6comp8_53_rct"
    g = golden[name]
    rc, inf, frame, err = _read(_host(), _ext(name), True, g["width"], g["height"], 4)
    assert rc == 0, err
    assert inf == [g["width"], g["height"], 4]
    pl = synth.planes(g["width"], g["height"], g["ncomp"], g["prec"], g["seed"], g["dist"])
    assert np.array_equal(frame.reshape(4, g["height"], g["width"]), pl[:4].astype(np.uint8))


# ------------------------------------------------------------------------------------------------ user-defined precincts: both directions on the GPU
PRECINCT_FILES = [n for n in EXT if n.startswith("p") or n in ("u3_300x200_rgb8_53_precincts_rpcl", "u4_300x200_rgb8_97_precincts_cprl_2layers",
                                                                "u8_300x200_rgb8_53_sop_eph_pcrl_precincts", "u9_256_rgb8_53_precincts_lrcp_tile100")]


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


STYLE_FILES = [n for n in EXT if n.startswith("s")] + ["u7_128_grey8_53_bypass_termall"]  # code-block styles: bypass, reset, termall, vcausal, pterm, segsym
CINEMA_FILES = [n for n in EXT if n.startswith("d")] + [n for n in EXT if n.startswith("r")]  # + region of interest (RGN, MAXSHIFT)  # libopenjp2's cinema profiles: tile-part per component, TLM, the 4K progression order change
MANY_COMPONENT_FILES = [n for n in EXT if n.startswith("m")]   # five / six components: the first four come out (src/common/j2k_openjpeg.cpp:278, :530)
SUPPORTED = CINEMA_FILES + PRECINCT_FILES + MANY_COMPONENT_FILES + ["u1_300x200_ycc420_8_53", "u2_301x199_ycc422_10_97_tile128", "u5_97x61_grey12_signed_53", "u6_200x150_rgb8_53_offset"] + STYLE_FILES


@pytest.mark.gpu
@pytest.mark.parametrize("name", SUPPORTED)
def test_files_outside_the_plug_ins_own_writer_decode_to_libopenjp2_samples(golden, name, opj):
    """Files libopenjp2 wrote with user-defined precincts (COD Scod bit 0) in every progression order, with SOP / EPH markers,
    tiles and layers; with 4:2:0 / 4:2:2 sub-sampled components; with a signed component; with every code-block style
    (selective bypass, context reset, termination on every pass, vertically causal contexts, predictable termination,
    segmentation symbols -- one by one and all at once): the GPU decode gives libopenjp2's
    samples at full and at half size.  A sub-sampled component is replicated onto the channel's full grid and a signed one
    offset by 2^(depth-1), as the reference's CopyChannel does (src/common/j2k_codec.cpp:250-252, :274, :374)."""
    g = golden[name]
    data = _ext(name)
    info = api.read_info(data)
    nout = min(g["ncomp"], 4)
    assert (info["width"], info["height"], info["channels"]) == (g["width"], g["height"], nout)
    e = api.Encoder(0)
    try:
        for red in (0, 1):
            dec = e.decode_planar(data, subsample=1 << red)
            ref = opj.decode_comps(data, red)
            h, w = -(-g["height"] >> red), -(-g["width"] >> red)
            assert dec.shape == (nout, h, w) and len(ref) == g["ncomp"]
            for c, (exp, comp) in enumerate(zip(g["decoded_comps"][str(red)], ref)):
                assert _sha(comp["data"]) == exp["sha256"]                      # the library on this box decodes what the committed hash says
                if c >= nout:
                    continue
                assert (info["sub_x"][c], info["sub_y"][c], info["comp_signed"][c]) == (comp["dx"], comp["dy"], comp["sgnd"])
                full = np.repeat(np.repeat(comp["data"], comp["dy"], axis=0), comp["dx"], axis=1)[:h, :w]
                if comp["sgnd"]:
                    full = full + (1 << (comp["prec"] - 1))
                assert np.array_equal(dec[c].astype(np.int32), full), (name, red, c)
    finally:
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in PRECINCT_FILES if "sop_eph" not in n])
def test_encoder_writes_libopenjp2_bytes_with_user_precincts(golden, name):
    """j2k_hip_params.num_precincts / precinct_w / precinct_h (OpenJPEG's res_spec semantics): the codestream is libopenjp2's,
    byte for byte -- COD with Scod bit 0 and the SPcod precinct bytes, packets in the progression's precinct order, the
    rate allocation pricing the same packets."""
    from j2k_amd import synth
    g = golden[name]
    kw = g["ext"]
    w, h, nc, prec = g["width"], g["height"], g["ncomp"], g["prec"]
    pl = synth.planes(w, h, nc, prec, g["seed"], "B")
    tile = kw.get("tile", (0, 0))
    assert tile[0] == tile[1]
    p = api.make_params(w, h, nc, prec, reversible=kw.get("reversible", True), ycc=kw.get("mct", False), num_resolutions=kw["numres"],
                        cblk=tuple(kw.get("cblk", (64, 64))), progression=kw.get("prog", 0), tile_size=tile[0], layers=kw.get("layers", 1),
                        rates=kw.get("rates"), precincts=[tuple(x) for x in kw["precincts"]], comment=g["comment"])
    e = api.Encoder(0)
    try:
        if nc in (3, 4):
            frame, lay = synth.ae_frame(pl, prec)
            got = e.encode_host(frame, lay, p)
        else:
            got = e.encode_planar_host(pl, p)
    finally:
        e.close()
    ref = _ext(name)
    assert len(got) == len(ref)
    assert got == ref


# ------------------------------------------------------------------------------------------------ QCC / COC (no encoder here writes them: crafted from libopenjp2's files)
def _marker(data: bytes, code: int):
    """(offset, payload) of the first marker segment `code` of the main header."""
    pos = 2
    while True:
        m = int.from_bytes(data[pos:pos + 2], "big")
        assert m != 0xFF90, hex(code)
        L = int.from_bytes(data[pos + 2:pos + 4], "big")
        if m == code:
            return pos, data[pos + 4:pos + 2 + L]
        pos += 2 + L


def _seg(code: int, payload: bytes) -> bytes:
    return code.to_bytes(2, "big") + (len(payload) + 2).to_bytes(2, "big") + payload


def _with_qcc(data: bytes, comps, wrong_qcd: bool, bump: int = 0) -> bytes:
    """`data` with a QCC segment for each of `comps` behind its QCD: the QCD's own values (exponents + `bump`).  wrong_qcd: the
    QCD itself is falsified (exponents + 1), so a reader that ignored the QCCs would decode something else."""
    pos, q = _marker(data, 0xFF5C)
    style = q[0] & 31

    def shifted(p: bytes, by: int) -> bytes:
        if by == 0:
            return p
        if style == 0:
            return p[:1] + bytes(((b >> 3) + by) << 3 for b in p[1:])
        vals = [int.from_bytes(p[1 + 2 * k:3 + 2 * k], "big") for k in range((len(p) - 1) // 2)]
        return p[:1] + b"".join(((v & 0x7FF) | (((v >> 11) + by) << 11)).to_bytes(2, "big") for v in vals)
    end = pos + 4 + len(q)
    qccs = b"".join(_seg(0xFF5D, bytes([c]) + shifted(q, bump)) for c in comps)
    return data[:pos] + _seg(0xFF5C, shifted(q, 1) if wrong_qcd else q) + qccs + data[end:]


def _with_coc(data: bytes, comps, levels_delta: int = 0) -> bytes:
    pos, cod = _marker(data, 0xFF52)
    sp = bytearray(cod[5:])           # levels, xcb, ycb, style, transform [, precincts]
    sp[0] += levels_delta
    end = pos + 4 + len(cod)
    cocs = b"".join(_seg(0xFF53, bytes([c, cod[0] & 1]) + bytes(sp)) for c in comps)
    return data[:end] + cocs + data[end:]


def test_qcc_and_redundant_coc_are_read_and_a_real_coc_goes_to_the_fallback():
    good = open(os.path.join(GOLDEN_DIR, "g6_300x200_rgb16_97_ict.j2k"), "rb").read()
    for data in (_with_qcc(good, (0, 1, 2), True), _with_qcc(good, (1,), False, 1), _with_coc(good, (0, 2))):
        i = api.read_info(data)
        assert (i["width"], i["height"], i["channels"]) == (300, 200, 3)
    with pytest.raises(api.J2kHipError) as ei:
        api.read_info(_with_coc(good, (1,), -1))
    assert ei.value.code == J2K_HIP_ERR_UNSUPPORTED and "COC" in str(ei.value)
    with pytest.raises(api.J2kHipError) as ei:
        api.read_info(_with_qcc(good, (3,), False))   # a component the image does not have: malformed, not unsupported
    assert ei.value.code == J2K_HIP_ERR_PARAM


@pytest.mark.gpu
@pytest.mark.parametrize("fname", ["g6_300x200_rgb16_97_ict.j2k", "g4_300x200_rgb16_53_rct_tile128.j2k", "q2_300x200_rgb8_97_ict_q30_38_45.j2k"])
def test_component_quantisation_overrides_decode_like_libopenjp2(fname, opj):
    """QCC: (1) every component carries the file's true quantisation in a QCC while the QCD lies -- the decode is the
    original's; (2) one component's QCC claims one more bit-plane per band -- whatever that means for the samples, it
    means the same to libopenjp2; (3) COC segments that repeat the COD change nothing."""
    good = open(os.path.join(GOLDEN_DIR, fname), "rb").read()
    e = api.Encoder(0)
    try:
        ref = e.decode_planar(good)
        assert np.array_equal(e.decode_planar(_with_qcc(good, (0, 1, 2), True)), ref)
        assert np.array_equal(e.decode_planar(_with_coc(good, (0, 1, 2))), ref)
        for comps, bump in (((1,), 1), ((0, 2), 1)):
            odd = _with_qcc(good, comps, False, bump)
            want = opj.decode_comps(odd)
            got = e.decode_planar(odd)
            for c, comp in enumerate(want):
                assert np.array_equal(got[c].astype(np.int64), np.clip(comp["data"], 0, (1 << comp["prec"]) - 1)), (fname, comps, c)
    finally:
        e.close()


def _with_palette(jp2: bytes, entries: int, columns: int, column_of=(0, 1, 2), depth: int = 8) -> tuple[bytes, np.ndarray]:
    """`jp2` (one component of indices) with a pclr + cmap pair added to its JP2 header: `entries` x `columns` palette values of
    `depth` bits, output channel i = palette column column_of[i] of component 0."""
    rng = np.random.default_rng(entries * 7 + columns)
    pal = rng.integers(0, 1 << depth, size=(entries, columns), dtype=np.int64)
    width = (depth + 7) // 8
    body = b"".join(int(v).to_bytes(width, "big") for v in pal.reshape(-1))
    pclr_payload = entries.to_bytes(2, "big") + bytes([columns]) + bytes([depth - 1] * columns) + body
    pclr = (8 + len(pclr_payload)).to_bytes(4, "big") + b"pclr" + pclr_payload
    cmap_payload = b"".join((0).to_bytes(2, "big") + bytes([1, c]) for c in column_of)
    cmap = (8 + len(cmap_payload)).to_bytes(4, "big") + b"cmap" + cmap_payload
    i = jp2.index(b"jp2h") - 4
    L = int.from_bytes(jp2[i:i + 4], "big")
    return jp2[:i] + (L + len(pclr) + len(cmap)).to_bytes(4, "big") + jp2[i + 4:i + L] + pclr + cmap + jp2[i + L:], pal


def test_palettised_jp2_reports_its_palette_like_the_reference(oracle, opj):
    """SURVEY 8f N4 / VERDICT r3 item 9: a JP2 file with a palette (pclr + cmap).  The reference's GetFileInfo fills FileInfo.LUT /
    LUTmap (src/common/j2k_openjpeg_codec.cpp:362-401) and its ReadFile decodes the INDEX component (OPJ_DPARAMETERS_IGNORE_PALETTE_FLAG,
    :503); the host applies the table.  j2k_hip_read_info reports the same table; applied to the decoded indices it gives the
    channels libopenjp2 produces when IT applies the palette."""
    jp2 = open(os.path.join(GOLDEN_DIR, "j2_64x48_grey8.jp2"), "rb").read()
    assert api.read_info(jp2)["lut_size"] == 0
    for column_of in ((0, 1, 2), (2, 0, 1)):
        crafted, pal = _with_palette(jp2, 256, 3, column_of)
        fi = api.read_info(crafted)
        assert fi["channels"] == 1 and fi["lut_size"] == 256 and fi["lut_channels"] == 3
        assert np.array_equal(np.array(fi["lut"]), pal) and tuple(fi["lut_column"][:3]) == column_of
        idx = oracle.decode(jp2)[0]  # (the codestream is the original's: the indices)
        assert idx.max() < 256
        ref = opj.decode_comps(crafted)  # as the reference drives it (ignore flag): the indices
        assert len(ref) == 1 and np.array_equal(ref[0]["data"], idx)
        if column_of != (0, 1, 2):
            continue  # (libopenjp2 2.4 refuses to apply a mapping whose columns are permuted: "Component 2 doesn't have a mapping")
        want = opj.decode_comps(crafted, apply_palette=True)  # libopenjp2 applying the palette itself: three components
        assert len(want) == 3
        for ch in range(3):
            assert np.array_equal(pal[idx, fi["lut_column"][ch]], want[ch]["data"]), (column_of, ch)


def test_palettes_beyond_the_references_limits_go_to_the_fallback():
    """What the reference's own GetFileInfo asserts against (more than 256 entries, other than three columns, columns deeper than
    8 bits, a channel that is not a palette column of component 0) is not decoded as if the indices were grey values: the file
    is the fallback's."""
    jp2 = open(os.path.join(GOLDEN_DIR, "j2_64x48_grey8.jp2"), "rb").read()
    direct = _with_palette(jp2, 16, 3)[0]
    direct = direct.replace(b"cmap" + (0).to_bytes(2, "big") + bytes([1, 0]), b"cmap" + (0).to_bytes(2, "big") + bytes([0, 0]), 1)  # MTYP 0: direct use
    for crafted in (_with_palette(jp2, 2, 1, (0,))[0], _with_palette(jp2, 300, 3)[0], _with_palette(jp2, 16, 3, depth=12)[0], direct):
        with pytest.raises(api.J2kHipError) as ei:
            api.read_info(crafted)
        assert ei.value.code == J2K_HIP_ERR_UNSUPPORTED and ("palett" in str(ei.value) or "mapping" in str(ei.value))


@pytest.mark.gpu
def test_palettised_jp2_decodes_to_its_indices():
    """j2k_hip_decode on a palettised file delivers the index component (what the reference's ReadFile asks OpenJPEG for)."""
    jp2 = open(os.path.join(GOLDEN_DIR, "j2_64x48_grey8.jp2"), "rb").read()
    crafted, _ = _with_palette(jp2, 256, 3)
    e = api.Encoder(0)
    try:
        assert np.array_equal(e.decode_planar(crafted), e.decode_planar(jp2))
    finally:
        e.close()


# ------------------------------------------------------------------------------------------------ packed packet headers (PPM / PPT): crafted from a SOP + EPH file
def _repack_headers(data: bytes, where: str) -> bytes:
    """The single-tile, single-tile-part SOP + EPH codestream `data` with every packet header (and its EPH) moved into PPT
    segments of the tile-part header (`where` = "ppt") or PPM segments of the main header ("ppm"); SOP markers and bodies stay.
    Neither codeword bytes nor packet-header bytes can contain a marker above 0xFF8F, so SOP / EPH are found by search."""
    sot = data.index(b"\xff\x90")
    psot = int.from_bytes(data[sot + 6:sot + 10], "big")
    sod = data.index(b"\xff\x93", sot)
    body, tail = data[sod + 2:sot + psot], data[sot + psot:]
    hdrs, rest, pos = b"", b"", 0
    while pos < len(body):
        assert body[pos:pos + 4] == b"\xff\x91\x00\x04", pos
        eph = body.index(b"\xff\x92", pos + 6)
        nxt = body.find(b"\xff\x91\x00\x04", eph + 2)
        nxt = len(body) if nxt < 0 else nxt
        hdrs += body[pos + 6:eph + 2]
        rest += body[pos:pos + 6] + body[eph + 2:nxt]
        pos = nxt

    def segs(code: int, payload: bytes) -> bytes:
        out = b""
        for z, k in enumerate(range(0, max(len(payload), 1), 60000)):
            chunk = payload[k:k + 60000]
            out += code.to_bytes(2, "big") + (len(chunk) + 3).to_bytes(2, "big") + bytes([z]) + chunk
        return out
    if where == "ppt":
        th = data[sot + 12:sod] + segs(0xFF61, hdrs)
        main = data[:sot]
    else:
        th = data[sot + 12:sod]
        main = data[:sot] + segs(0xFF60, len(hdrs).to_bytes(4, "big") + hdrs)
    new_psot = 12 + len(th) + 2 + len(rest)
    return main + data[sot:sot + 6] + new_psot.to_bytes(4, "big") + data[sot + 10:sot + 12] + th + b"\xff\x93" + rest + tail


PACKED_SOURCE = "u8_300x200_rgb8_53_sop_eph_pcrl_precincts"


@pytest.mark.parametrize("where", ["ppt", "ppm"])
def test_packed_packet_headers_are_read(where, opj):
    """PPT / PPM files parse (no encoder here writes them: the packet headers of a SOP + EPH file are moved into the marker
    segments); libopenjp2 reads the crafted file as it reads the original -- the GPU test holds this decoder to the same."""
    good = _ext(PACKED_SOURCE)
    crafted = _repack_headers(good, where)
    assert len(crafted) > len(good) and (b"\xff\x61" if where == "ppt" else b"\xff\x60") in crafted
    i = api.read_info(crafted)
    assert (i["width"], i["height"], i["channels"]) == (300, 200, 3)
    a, b = opj.decode_comps(good), opj.decode_comps(crafted)
    assert all(np.array_equal(x["data"], y["data"]) for x, y in zip(a, b))


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["ppt", "ppm"])
def test_packed_packet_headers_decode_like_the_original(where):
    good = _ext(PACKED_SOURCE)
    e = api.Encoder(0)
    try:
        for sub in (1, 2):
            assert np.array_equal(e.decode_planar(_repack_headers(good, where), subsample=sub), e.decode_planar(good, subsample=sub))
    finally:
        e.close()

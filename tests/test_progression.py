"""Progression orders (settings.order; the reference stores it, its WriteFile never passes it on:
src/common/j2k_openjpeg_codec.cpp:703-709): oracle and GPU path against libopenjp2 for RLCP, RPCL, PCRL, CPRL."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from j2k_amd import api, synth
from oracle.oracle import make_params

ORDERS = ["o1_300x200_rgb8_53_rct_3layers_rlcp", "o2_300x200_rgb8_97_ict_rpcl_r40_20_8",
          "o3_200x300_rgba16_53_tile128_2layers_pcrl", "o4_97x61_grey12_97_cprl_r12_3"]


def case(golden, name):
    g = golden[name]
    pl = synth.planes(g["width"], g["height"], g["ncomp"], g["prec"], g["seed"], g["dist"])
    f = open(os.path.join(GOLDEN_DIR, name + ".j2k"), "rb").read()
    assert hashlib.sha256(f).hexdigest() == g["sha256"]
    return g, pl, f


@pytest.mark.parametrize("name", ORDERS)
def test_oracle_progression_matches_golden(oracle, golden, name):
    g, pl, f = case(golden, name)
    p = make_params(g["width"], g["height"], g["ncomp"], g["prec"], prog=g["order"], **g["params"])
    got = oracle.encode_rates(pl, p, g["rates"], comment=g["comment"]) if g["rates"] else oracle.encode(pl, p, comment=g["comment"])
    assert got == f


def test_main_header_carries_the_order():
    for order in range(5):
        h = api.main_header(api.make_params(64, 64, 1, 8, num_resolutions=2, progression=order))
        cod = h.index(b"\xff\x52")
        assert h[cod + 5] == order
    with pytest.raises(api.J2kHipError):
        api.main_header(api.make_params(64, 64, 1, 8, num_resolutions=2, progression=5))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ORDERS)
def test_gpu_progression_matches_golden(golden, name):
    g, pl, f = case(golden, name)
    kw = g["params"]
    p = api.make_params(g["width"], g["height"], g["ncomp"], g["prec"], reversible=kw.get("reversible", True),
                        ycc=kw.get("mct", False), layers=kw.get("layers", 1), tile_size=kw.get("tile", 0),
                        num_resolutions=kw.get("numres", 6), comment=g["comment"], rates=g["rates"], progression=g["order"])
    enc = api.Encoder(0)
    if g["ncomp"] in (3, 4):
        frame, lay = synth.ae_frame(pl, g["prec"])
        got = enc.encode_host(frame, lay, p)
    else:
        got = enc.encode_planar_host(pl, p)
    enc.close()
    assert got == f


@pytest.mark.gpu
def test_hip_codec_honours_settings_order(golden, monkeypatch):
    """HipCodec::HonourSettings passes settings.order on; ReferenceLiteral keeps LRCP like the reference."""
    from oracle.oracle import strip_com
    api.load_library()
    H = C.CDLL(os.path.join(os.path.dirname(api.LIBPATH), "libj2k_host.so"))
    H.j2k_host_test_write.restype = C.c_long
    H.j2k_host_test_write.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_ulong, C.c_char_p, C.c_ulong]
    g, pl, f = case(golden, "o1_300x200_rgb8_53_rct_3layers_rlcp")
    frame, lay = synth.ae_frame(pl, 8)
    out = np.empty(1 << 22, dtype=np.uint8)
    err = C.create_string_buffer(512)
    monkeypatch.setenv("J2K_HOST_TEST_ORDER", "1")

    def write(honour):
        n = H.j2k_host_test_write(frame.ctypes.data, 300, 200, lay["rowbytes"], lay["sample_bytes"], 3, 8, 1, 1, 3, 0, int(honour), -1,
                                  out.ctypes.data, out.nbytes, err, 512)
        assert n > 0, err.value
        return out[:n].tobytes()
    assert strip_com(write(True)) == strip_com(f)
    lit = write(False)
    assert lit[lit.index(b"\xff\x52") + 5] == 0  # LRCP

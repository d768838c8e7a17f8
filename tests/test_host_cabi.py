"""CPU: the C-ABI library loads, exports every symbol include/j2k_hip.h declares, validates
parameters like the reference path does, and refuses to run without a HIP device (no CPU fallback).
No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from j2k_amd import api


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "j2k_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(j2k_hip_[a-z0-9_]+)\s*\(", txt)) - {"j2k_hip_write_fn"})


def test_library_exports_every_declared_symbol():
    L = api.load_library()
    syms = _declared_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(L, s), s
    assert sorted(api.EXPORTS) == syms
    assert L.j2k_hip_abi_version() == 9


@pytest.mark.parametrize("name", ["g8_c1", "g8_c2", "g8_c3", "g8_c3_5lvl", "g8_c4", "g8_c5"])
def test_main_header_matches_golden(golden, name):
    g = golden[name]
    kw = g["params"]
    p = api.make_params(g["width"], g["height"], g["ncomp"], g["prec"], reversible=kw.get("reversible", True),
                        ycc=kw.get("mct", False), tile_size=kw.get("tile", 0), num_resolutions=kw.get("numres", 6))
    L = api.load_library()
    buf = np.empty(1024, dtype=np.uint8)
    n, nt = C.c_size_t(), C.c_uint32()
    assert L.j2k_hip_main_header(C.byref(p), buf.ctypes.data, 1024, C.byref(n), C.byref(nt)) == 0
    assert buf[:n.value].tobytes().hex() == g["main_header_hex"]


def test_comment_segment():
    L = api.load_library()
    buf = np.empty(1024, dtype=np.uint8)
    n, nt = C.c_size_t(), C.c_uint32()
    p = api.make_params(64, 64, 1, 8, comment="Created by OpenJPEG version 2.4.0")
    assert L.j2k_hip_main_header(C.byref(p), buf.ctypes.data, 1024, C.byref(n), C.byref(nt)) == 0
    h = buf[:n.value].tobytes()
    assert h.endswith(b"\xff\x64\x00\x25\x00\x01Created by OpenJPEG version 2.4.0")
    p = api.make_params(64, 64, 1, 8, comment=None)
    assert L.j2k_hip_main_header(C.byref(p), buf.ctypes.data, 1024, C.byref(n), C.byref(nt)) == 0
    assert b"Created by j2k_hip" in buf[:n.value].tobytes()


BAD = [
    dict(width=0, height=8, channels=1, depth=8),
    dict(width=8, height=8, channels=5, depth=8),
    dict(width=8, height=8, channels=1, depth=17),
    dict(width=64, height=64, channels=1, depth=8, ycc=True),            # MCT needs 3 components
    dict(width=64, height=64, channels=1, depth=8, num_resolutions=8),   # too many resolutions for the tile
    dict(width=64, height=64, channels=1, depth=8, cblk=(128, 32)),
    dict(width=64, height=64, channels=1, depth=8, cblk=(48, 64)),
    dict(width=300, height=200, channels=3, depth=8, tile_size=16),      # tile smaller than 2^(numres-1)
]


@pytest.mark.parametrize("kw", BAD, ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
def test_parameter_validation(kw):
    L = api.load_library()
    p = api.make_params(**kw)
    n, nt = C.c_size_t(), C.c_uint32()
    assert L.j2k_hip_main_header(C.byref(p), None, 0, C.byref(n), C.byref(nt)) == 1  # J2K_HIP_ERR_PARAM


def test_struct_size_guard():
    L = api.load_library()
    p = api.make_params(64, 64, 1, 8)
    p.struct_size = 12
    n, nt = C.c_size_t(), C.c_uint32()
    assert L.j2k_hip_main_header(C.byref(p), None, 0, C.byref(n), C.byref(nt)) == 1


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.J2kHipError) as ei:
        api.Encoder(0)
    assert ei.value.code == 2  # J2K_HIP_ERR_DEVICE


def test_host_codec_library_loads():
    from j2k_amd import build
    path = os.path.join(os.path.dirname(api.LIBPATH), "libj2k_host.so")
    if not os.path.exists(path):
        build.build_host()
    api.load_library()
    H = C.CDLL(path)
    H.j2k_host_codec_name.restype = C.c_char_p
    assert H.j2k_host_codec_name() == b"HIP"  # sorts before "OpenJPEG" -> becomes the default codec
    assert hasattr(H, "j2k_host_test_write")


REFERENCE_COMMON = "/root/reference/src/common"


@pytest.mark.skipif(not os.path.isdir(REFERENCE_COMMON), reason="the plug-in's headers are not present on this machine")
def test_hip_codec_compiles_against_the_plugin_headers(tmp_path):
    """hip_codec.cpp built the way a maintainer would build it inside the plug-in tree: with the plug-in's own
    j2k_codec.h instead of our re-declaration (compile only -- nothing of the reference is copied or linked)."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    obj = str(tmp_path / "hip_codec.o")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-c", "-DJ2K_HIP_USE_PLUGIN_HEADERS", "-I" + REFERENCE_COMMON,
                        "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "j2k_amd", "host", "hip_codec.cpp"), "-o", obj],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert os.path.getsize(obj) > 0


@pytest.mark.skipif(not os.path.isdir(REFERENCE_COMMON), reason="the plug-in's headers are not present on this machine")
def test_redeclared_interface_matches_the_plugin_header():
    """j2k_codec_api.h re-declares the plug-in's enums and limits so that hip_codec.cpp builds outside the tree;
    their order and values must stay those of the plug-in's j2k_codec.h (read as text, nothing copied)."""
    ref = open(os.path.join(REFERENCE_COMMON, "j2k_codec.h")).read()
    mine = open(os.path.join(ROOT, "j2k_amd", "host", "j2k_codec_api.h")).read()

    def enums(t):
        t = re.sub(r"//.*", "", t)
        t = re.sub(r"/\*.*?\*/", "", t, flags=re.S)
        return {m.group(1): [re.sub(r"\s*=.*", "", x.strip()) for x in m.group(2).split(",") if x.strip()]
                for m in re.finditer(r"enum\s+(\w+)\s*\{([^}]*)\}", t)}
    a, b = enums(ref), enums(mine)
    assert a, "no enums found in the plug-in header"
    for name, members in a.items():
        assert b.get(name) == members, name
    for d in ("J2K_CODEC_MAX_CHANNELS", "J2K_CODEC_MAX_LUT_ENTRIES", "J2K_CODEC_MAX_LAYERS"):
        assert re.search(r"#define\s+%s\s+(\d+)" % d, ref).group(1) == re.search(r"#define\s+%s\s+(\d+)" % d, mine).group(1), d


def test_native_sinks_copy_and_count():
    """j2k_hip_debug_copy_sink / _count_sink (bench.py's host_path sinks): plain C functions with the write callback's
    signature -- appended bytes, a refused overflow (0 = what the encoder reports as a sink error), a running count."""
    import ctypes as C
    import numpy as np
    L = api.load_library()
    L.j2k_hip_debug_copy_sink.restype = C.c_size_t
    L.j2k_hip_debug_copy_sink.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.j2k_hip_debug_count_sink.restype = C.c_size_t
    L.j2k_hip_debug_count_sink.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    dst = np.zeros(10, dtype=np.uint8)
    st = api.CopySink(dst.ctypes.data, 10, 0)
    a, b = np.arange(6, dtype=np.uint8), np.arange(6, 12, dtype=np.uint8)
    assert L.j2k_hip_debug_copy_sink(C.byref(st), a.ctypes.data, 6) == 6 and st.pos == 6
    assert L.j2k_hip_debug_copy_sink(C.byref(st), b.ctypes.data, 6) == 0 and st.pos == 6   # would overflow: refused whole
    assert L.j2k_hip_debug_copy_sink(C.byref(st), b.ctypes.data, 4) == 4 and st.pos == 10
    assert dst.tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 8, 9]
    n = C.c_size_t(0)
    assert L.j2k_hip_debug_count_sink(C.byref(n), a.ctypes.data, 6) == 6 and L.j2k_hip_debug_count_sink(C.byref(n), None, 5) == 5 and n.value == 11
    fn = api.native_sink(L, True)   # the same function as a j2k_hip_write_fn value
    st.pos = 0
    assert fn(C.cast(C.pointer(st), C.c_void_p), a.ctypes.data, 3) == 3 and st.pos == 3


def test_cinema_profile_main_header_and_parameter_checks():
    """dci_profile on the CPU: the main header j2k_hip_main_header writes for Rsiz 3 / 4 (COD overridden by the profile, TLM with
    one entry per tile-part, the 4K progression order change, no tiles) and the parameter sets normalise refuses."""
    p = api.make_params(2048, 1080, 3, 12, reversible=True, ycc=False, layers=5, tile_size=512, num_resolutions=9, cblk=(64, 64), dci_profile=3, comment="")
    h = api.main_header(p)
    assert h[6:8] == b"\x00\x03"
    cod = h.index(b"\xff\x52")
    assert h[cod + 4] == 1 and h[cod + 5] == 4 and h[cod + 6:cod + 8] == b"\x00\x01" and h[cod + 8] == 1        # precincts, CPRL, 1 layer, MCT
    assert h[cod + 9] == 5 and h[cod + 10:cod + 12] == b"\x03\x03" and h[cod + 12] == 0 and h[cod + 13] == 0    # 6 resolutions, 32 x 32, 9/7
    assert h[cod + 14:cod + 20] == b"\x77\x88\x88\x88\x88\x88"
    assert int.from_bytes(h[24:28], "big") == 2048 and int.from_bytes(h[28:32], "big") == 1080                  # one tile = the image
    tlm = h.index(b"\xff\x55")
    assert int.from_bytes(h[tlm + 2:tlm + 4], "big") == 4 + 5 * 3 and h[tlm + 5] == 0x50 and b"\xff\x5f" not in h and b"\xff\x64" not in h
    h4 = api.main_header(api.make_params(4096, 2160, 3, 12, num_resolutions=7, dci_profile=4, comment=""))
    assert h4[6:8] == b"\x00\x04"
    tlm = h4.index(b"\xff\x55")
    assert int.from_bytes(h4[tlm + 2:tlm + 4], "big") == 4 + 5 * 6
    poc = h4.index(b"\xff\x5f")
    assert h4[poc + 2:poc + 18] == bytes([0, 16, 0, 0, 0, 1, 6, 3, 4, 6, 0, 0, 1, 7, 3, 4])
    for bad in (dict(channels=4), dict(depth=10), dict(width=2049), dict(height=1081), dict(profile=5), dict(rates=[20.0])):
        kw = dict(width=2048, height=1080, channels=3, depth=12, profile=3, rates=None)
        kw.update(bad)
        with pytest.raises(api.J2kHipError):
            api.main_header(api.make_params(kw["width"], kw["height"], kw["channels"], kw["depth"], dci_profile=kw["profile"], rates=kw["rates"]))
    with pytest.raises(api.J2kHipError):
        api.main_header(api.make_params(4096, 2160, 3, 12, num_resolutions=1, dci_profile=4))

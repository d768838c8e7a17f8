"""ctypes front-ends for the TEST-ONLY checkers in oracle/.

* ``Oracle``   -- our plain-C restatement (oracle/j2k_oracle.c -> libj2k_oracle.so).
* ``OpjReplay`` -- replay of the reference's OpenJPEG call sequence
  (reference: src/common/j2k_openjpeg_codec.cpp:598-750) against a libopenjp2 found at run time
  (oracle/opj_replay.c -> _ref/libopj_replay.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Nothing here reads /root/reference.
"""
from __future__ import annotations

import ctypes as C
import glob
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "width", "height", "ncomp", "prec", "reversible", "mct", "numres",
        "cblkw_exp", "cblkh_exp", "layers", "tile_w", "tile_h", "prog")]


def make_params(width, height, ncomp, prec, reversible=True, mct=False, numres=6,
                cblk=(64, 64), layers=1, tile=0, prog=0):
    tw, th = (tile, tile) if isinstance(tile, int) else tile
    return Params(width, height, ncomp, prec, int(reversible), int(mct), numres,
                  cblk[0].bit_length() - 1, cblk[1].bit_length() - 1, layers, tw, th, prog)


def build(force: bool = False) -> None:
    """Compile the checkers (gcc only). Building the checker is not using it."""
    need = force or not os.path.exists(os.path.join(HERE, "libj2k_oracle.so"))
    src_m = max(os.path.getmtime(os.path.join(HERE, f)) for f in ("j2k_oracle.c", "j2k_oracle_dec.c", "j2k_oracle.h"))
    if not need and os.path.getmtime(os.path.join(HERE, "libj2k_oracle.so")) < src_m:
        need = True
    if need:
        subprocess.check_call(["make", "-C", HERE, "libj2k_oracle.so"], stdout=subprocess.DEVNULL)
    ref = os.path.join(HERE, "_ref", "libopj_replay.so")
    if force or not os.path.exists(ref) or os.path.getmtime(ref) < os.path.getmtime(os.path.join(HERE, "opj_replay.c")):
        subprocess.call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def _i32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


class Oracle:
    def __init__(self):
        build()
        L = C.CDLL(os.path.join(HERE, "libj2k_oracle.so"))
        L.j2ko_encode_ex.restype = C.c_long
        L.j2ko_encode_ex.argtypes = [C.POINTER(Params), C.POINTER(C.c_int32), C.POINTER(C.c_uint8),
                                     C.c_size_t, C.c_char_p, C.POINTER(C.c_int32)]
        L.j2ko_t1_encode_block.restype = C.c_int
        L.j2ko_t1_encode_block.argtypes = [C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int,
                                           C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_int),
                                           C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint8),
                                           C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        L.j2ko_included_passes.restype = C.c_int
        L.j2ko_included_passes.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.j2ko_copy_channel.restype = None
        L.j2ko_copy_channel.argtypes = [C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_void_p, C.c_ssize_t,
                                        C.c_ssize_t, C.c_int, C.c_int, C.c_int]
        L.j2ko_dwt53.restype = None
        L.j2ko_dwt53.argtypes = [C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.j2ko_dwt97.restype = None
        L.j2ko_dwt97.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.j2ko_band_quant.restype = None
        L.j2ko_band_quant.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int), C.POINTER(C.c_float)]
        L.j2ko_promote.restype = C.c_uint16
        L.j2ko_promote.argtypes = [C.c_uint16]
        L.j2ko_demote.restype = C.c_uint16
        L.j2ko_demote.argtypes = [C.c_uint16]
        L.j2ko_encode_rates.restype = C.c_long
        L.j2ko_encode_rates.argtypes = [C.POINTER(Params), C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.c_size_t,
                                        C.c_char_p, C.POINTER(C.c_float)]
        L.j2ko_encode_psnr.restype = C.c_long
        L.j2ko_encode_psnr.argtypes = L.j2ko_encode_rates.argtypes
        L.j2ko_encode_rates_ex.restype = C.c_long
        L.j2ko_encode_rates_ex.argtypes = L.j2ko_encode_rates.argtypes + [C.c_size_t]
        L.j2ko_jp2_header.restype = C.c_size_t
        L.j2ko_jp2_header.argtypes = [C.c_uint32] * 4 + [C.c_int, C.c_char_p, C.c_uint32, C.c_int, C.c_uint32,
                                                        C.POINTER(C.c_uint8), C.c_size_t]
        L.j2ko_quant97.restype = C.c_int32
        L.j2ko_quant97.argtypes = [C.c_float, C.c_float]
        L.j2ko_decode.restype = C.c_int
        L.j2ko_decode.argtypes = [C.POINTER(C.c_uint8), C.c_size_t, C.c_int, C.POINTER(C.c_int32), C.c_size_t, C.POINTER(C.c_int)]
        L.j2ko_decode_error.restype = C.c_char_p
        L.j2ko_decode_info.restype = C.c_int
        L.j2ko_decode_info.argtypes = [C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_int)]
        L.j2ko_t1_decode_block.restype = C.c_int
        L.j2ko_t1_decode_block.argtypes = [C.POINTER(C.c_uint8), C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)]
        L.j2ko_idwt53.restype = None
        L.j2ko_idwt53.argtypes = L.j2ko_dwt53.argtypes
        L.j2ko_idwt97.restype = None
        L.j2ko_idwt97.argtypes = L.j2ko_dwt97.argtypes
        L.j2ko_copy_channel_out.restype = None
        L.j2ko_copy_channel_out.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int,
                                            C.POINTER(C.c_int32), C.c_int, C.c_int]
        self.L = L

    # -- decode path ------------------------------------------------------------------------------
    def decode_info(self, data: bytes) -> dict:
        buf = np.frombuffer(data, dtype=np.uint8)
        info = (C.c_int * 12)()
        if self.L.j2ko_decode_info(_u8p(buf), len(data), info) != 0:
            raise RuntimeError("oracle decode failed: " + self.L.j2ko_decode_error().decode())
        keys = ("width", "height", "ncomp", "prec", "reversible", "mct", "numres", "jp2", "enumcs", "icc_off", "icc_len", "alpha_mask")
        return dict(zip(keys, list(info)))

    def decode(self, data: bytes, reduce: int = 0) -> np.ndarray:
        """(ncomp, h, w) int32 samples of the image at resolution `reduce`."""
        i = self.decode_info(data)
        w, h = -(-i["width"] >> reduce), -(-i["height"] >> reduce)
        buf = np.frombuffer(data, dtype=np.uint8)
        out = np.empty((i["ncomp"], h, w), dtype=np.int32)
        dims = (C.c_int * 4)()
        if self.L.j2ko_decode(_u8p(buf), len(data), reduce, _i32p(out), out.size, dims) != 0:
            raise RuntimeError("oracle decode failed: " + self.L.j2ko_decode_error().decode())
        assert (dims[0], dims[1], dims[2]) == (w, h, i["ncomp"])
        return out

    def idwt53(self, plane: np.ndarray, levels: int, x0: int = 0, y0: int = 0):
        a = np.ascontiguousarray(plane, dtype=np.int32).copy()
        self.L.j2ko_idwt53(_i32p(a), a.shape[1], a.shape[0], a.shape[1], x0, y0, levels)
        return a

    def idwt97(self, plane: np.ndarray, levels: int, x0: int = 0, y0: int = 0):
        a = np.ascontiguousarray(plane, dtype=np.float32).copy()
        self.L.j2ko_idwt97(a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[1], a.shape[0], a.shape[1], x0, y0, levels)
        return a

    def t1_decode_block(self, data: bytes, w: int, h: int, orient: int, numbps: int, npasses: int) -> np.ndarray:
        buf = np.frombuffer(data + b"\0", dtype=np.uint8)
        out = np.empty((h, w), dtype=np.int32)
        self.L.j2ko_t1_decode_block(_u8p(buf), len(data), w, h, orient, numbps, npasses, _i32p(out))
        return out

    def copy_channel_out(self, src: np.ndarray, src_depth: int, dst_bytes: int, dst_depth: int, colbytes: int, rowbytes: int,
                         width: int, height: int) -> np.ndarray:
        """src: (h, w) int32 plane -> a buffer of height*rowbytes bytes holding the strided channel."""
        src = np.ascontiguousarray(src, dtype=np.int32)
        dst = np.zeros(height * rowbytes, dtype=np.uint8)
        self.L.j2ko_copy_channel_out(dst.ctypes.data, dst_bytes, dst_depth, colbytes, rowbytes, width, height, _i32p(src),
                                     src.shape[1], src_depth)
        return dst

    # -- whole path -----------------------------------------------------------------------------
    def encode(self, planes: np.ndarray, params: Params, comment: str | None = None,
               want_coefs: bool = False):
        """planes: (ncomp, h, w) int32 of unsigned sample values (post-CopyBuffer)."""
        planes = np.ascontiguousarray(planes, dtype=np.int32)
        assert planes.shape == (params.ncomp, params.height, params.width)
        cap = planes.size * 4 + (1 << 20)
        out = np.empty(cap, dtype=np.uint8)
        coefs = np.empty(planes.shape, dtype=np.int32) if want_coefs else None
        n = self.L.j2ko_encode_ex(C.byref(params), _i32p(planes), _u8p(out), cap,
                                  comment.encode() if comment is not None else None,
                                  _i32p(coefs) if want_coefs else None)
        if n < 0:
            raise RuntimeError(f"oracle encode failed: {n}")
        cs = out[:n].tobytes()
        return (cs, coefs) if want_coefs else cs

    def encode_rates(self, planes: np.ndarray, params: Params, rates, comment: str | None = None, prefix_len: int = 0) -> bytes:
        """Rate-controlled encode; params.layers must equal len(rates).  prefix_len = bytes of a file wrapper
        in front of the codestream (they count against the budget)."""
        planes = np.ascontiguousarray(planes, dtype=np.int32)
        assert planes.shape == (params.ncomp, params.height, params.width) and params.layers == len(rates)
        cap = planes.size * 4 + (1 << 20)
        out = np.empty(cap, dtype=np.uint8)
        r = (C.c_float * len(rates))(*rates)
        n = self.L.j2ko_encode_rates_ex(C.byref(params), _i32p(planes), _u8p(out), cap,
                                        comment.encode() if comment is not None else None, r, prefix_len)
        if n < 0:
            raise RuntimeError(f"oracle encode failed: {n}")
        return out[:n].tobytes()

    def encode_psnr(self, planes: np.ndarray, params: Params, psnr, comment: str | None = None) -> bytes:
        """Fixed-quality encode (PSNR target per layer); params.layers must equal len(psnr)."""
        planes = np.ascontiguousarray(planes, dtype=np.int32)
        assert planes.shape == (params.ncomp, params.height, params.width) and params.layers == len(psnr)
        cap = planes.size * 4 + (1 << 20)
        out = np.empty(cap, dtype=np.uint8)
        r = (C.c_float * len(psnr))(*psnr)
        n = self.L.j2ko_encode_psnr(C.byref(params), _i32p(planes), _u8p(out), cap,
                                    comment.encode() if comment is not None else None, r)
        if n < 0:
            raise RuntimeError(f"oracle encode failed: {n}")
        return out[:n].tobytes()

    def jp2_wrap(self, codestream: bytes, params: Params, color_space: int = 0, icc: bytes | None = None,
                 alpha_channel: int = -1) -> bytes:
        """JP2 file = box prefix (j2ko_jp2_header) + codestream."""
        buf = np.empty(4096 + (len(icc) if icc else 0), dtype=np.uint8)
        n = self.L.j2ko_jp2_header(params.width, params.height, params.ncomp, params.prec, color_space,
                                   icc if icc else None, len(icc) if icc else 0, alpha_channel, len(codestream),
                                   _u8p(buf), buf.size)
        return buf[:n].tobytes() + codestream

    # -- stages ---------------------------------------------------------------------------------
    def copy_channel(self, src: np.ndarray, base_off: int, width, height, colbytes, rowbytes,
                     src_bytes, src_depth, dst_depth):
        dst = np.empty((height, width), dtype=np.int32)
        self.L.j2ko_copy_channel(_i32p(dst), width, height, src.ctypes.data + base_off, colbytes,
                                 rowbytes, src_bytes, src_depth, dst_depth)
        return dst

    def dwt53(self, plane: np.ndarray, levels: int, x0: int = 0, y0: int = 0):
        a = np.ascontiguousarray(plane, dtype=np.int32).copy()
        h, w = a.shape
        self.L.j2ko_dwt53(_i32p(a), w, h, w, x0, y0, levels)
        return a

    def dwt97(self, plane: np.ndarray, levels: int, x0: int = 0, y0: int = 0):
        a = np.ascontiguousarray(plane, dtype=np.float32).copy()
        h, w = a.shape
        self.L.j2ko_dwt97(a.ctypes.data_as(C.POINTER(C.c_float)), w, h, w, x0, y0, levels)
        return a

    def band_quant(self, prec, reversible, numres, bandidx):
        e, m, nb, ss = C.c_int(), C.c_int(), C.c_int(), C.c_float()
        self.L.j2ko_band_quant(prec, int(reversible), numres, bandidx, C.byref(e), C.byref(m),
                               C.byref(nb), C.byref(ss))
        return e.value, m.value, nb.value, ss.value

    def t1_block(self, data: np.ndarray, orient: int, want_symbols: bool = False):
        """data: (h, w) int32 already scaled (6 fractional bits)."""
        data = np.ascontiguousarray(data, dtype=np.int32)
        h, w = data.shape
        cap = w * h * 4 + 64
        out = np.empty(cap, dtype=np.uint8)
        rate = (C.c_int * 128)()
        nms = (C.c_int * 128)()
        pns = (C.c_int * 128)()
        nb = C.c_int()
        ns = C.c_size_t()
        symcap = w * h * 40 if want_symbols else 0
        sym = np.empty(max(symcap, 1), dtype=np.uint8)
        np_ = self.L.j2ko_t1_encode_block(_i32p(data), w, h, orient, _u8p(out), cap, C.byref(nb), rate, nms,
                                          _u8p(sym) if want_symbols else None, symcap, C.byref(ns), pns)
        if np_ < 0:
            raise RuntimeError("t1 overflow")
        incl = self.L.j2ko_included_passes(np_, rate, nms)
        total = rate[np_ - 1] if np_ else 0
        res = dict(numbps=nb.value, npasses=np_, rates=list(rate[:np_]), nmsedec=list(nms[:np_]),
                   included=incl, length=(rate[incl - 1] if incl else 0), data=out[:total].tobytes())
        if want_symbols:
            assert ns.value <= symcap
            res["symbols"] = sym[:ns.value].copy()
            res["pass_nsym"] = list(pns[:np_])
        return res


def find_openjpeg_libs():
    """Candidate libopenjp2 binaries, most trusted first ($J2K_OPJ_LIB, conda, system, Pillow)."""
    cands = []
    env = os.environ.get("J2K_OPJ_LIB")
    if env:
        cands.append(env)
    pats = ["/opt/conda/lib/libopenjp2.so.2.*", "/usr/lib/x86_64-linux-gnu/libopenjp2.so.2.*",
            "/usr/lib/libopenjp2.so.2.*", "/usr/local/lib/libopenjp2.so.2.*"]
    try:
        import PIL  # noqa: F401  (only used to locate its bundled library)
        pats.append(os.path.join(os.path.dirname(os.path.dirname(PIL.__file__)), "pillow.libs", "libopenjp2*.so*"))
    except Exception:
        pass
    for p in pats:
        cands.extend(sorted(glob.glob(p)))
    seen, out = set(), []
    for c in cands:
        r = os.path.realpath(c)
        if os.path.exists(r) and r not in seen:
            seen.add(r)
            out.append(r)
    return out


class OpjReplay:
    """Drives a real libopenjp2 exactly like OpenJPEGCodec::WriteFile does."""

    def __init__(self, libpath: str | None = None):
        build()
        so = os.path.join(HERE, "_ref", "libopj_replay.so")
        if not os.path.exists(so):
            raise OSError("oracle/_ref/libopj_replay.so not built (openjpeg.h missing at build time)")
        L = C.CDLL(so)
        L.opjr_open.argtypes = [C.c_char_p]
        L.opjr_version.restype = C.c_char_p
        L.opjr_libpath.restype = C.c_char_p
        L.opjr_last_error.restype = C.c_char_p
        L.opjr_encode.restype = C.c_long
        L.opjr_encode.argtypes = [C.POINTER(C.c_int32)] + [C.c_int] * 13 + [C.POINTER(C.c_uint8), C.c_size_t,
                                                                          C.POINTER(C.c_double)]
        L.opjr_encode_jp2.restype = C.c_long
        L.opjr_encode_jp2.argtypes = [C.POINTER(C.c_int32)] + [C.c_int] * 14 + [C.c_void_p, C.c_uint32, C.c_int,
                                                                              C.POINTER(C.c_uint8), C.c_size_t,
                                                                              C.POINTER(C.c_double)]
        L.opjr_encode_jp2_rates.restype = C.c_long
        L.opjr_encode_jp2_rates.argtypes = [C.POINTER(C.c_int32)] + [C.c_int] * 10 + [C.POINTER(C.c_float)] + [C.c_int] * 4 + \
                                           [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_double)]
        L.opjr_encode_psnr.restype = C.c_long
        L.opjr_encode_psnr.argtypes = [C.POINTER(C.c_int32)] + [C.c_int] * 10 + [C.POINTER(C.c_float)] + [C.c_int] * 3 + \
                                      [C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_double)]
        L.opjr_encode_rates.restype = C.c_long
        L.opjr_encode_rates.argtypes = [C.POINTER(C.c_int32)] + [C.c_int] * 10 + [C.POINTER(C.c_float)] + [C.c_int] * 3 + \
                                       [C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_double)]
        L.opjr_set_progression.argtypes = [C.c_int]
        L.opjr_set_progression.restype = None
        L.opjr_decode_ex.restype = C.c_int
        L.opjr_decode_ex.argtypes = [C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_int32), C.c_size_t,
                                     C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint8), C.c_size_t]
        L.opjr_decode.restype = C.c_int
        L.opjr_decode.argtypes = [C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_int32), C.c_size_t,
                                  C.POINTER(C.c_int), C.c_int]
        self.L = L
        paths = [libpath] if libpath else find_openjpeg_libs()
        for p in paths:
            if L.opjr_open(p.encode()) == 0:
                break
        else:
            raise OSError("no usable libopenjp2 found")
        self.version = L.opjr_version().decode()
        self.libpath = L.opjr_libpath().decode()
        OpjReplay._active = self.libpath
        self.last_seconds = 0.0

    _active = None  # the library the (process-wide) replay object has open: every instance re-opens its own before a call

    def _sel(self):
        if OpjReplay._active != self.libpath:
            if self.L.opjr_open(self.libpath.encode()) != 0:
                raise OSError("cannot re-open " + self.libpath)
            OpjReplay._active = self.libpath

    def encode_ext(self, comps, x0=0, y0=0, x1=None, y1=None, sub=None, prec=8, sgnd=None, reversible=True, mct=False, numres=6,
                   cblk=(64, 64), layers=1, tile=(0, 0), tile_origin=(0, 0), prog=0, sop=False, eph=False, mode=0, precincts=None,
                   rsiz=0, max_cs_size=0, max_comp_size=0, rates=None, threads=0, roi=None) -> bytes:
        """General encode (opjr_encode_ext): comps = list of 2-D int32 arrays, one per component, each of the size its
        sub-sampling factors sub[c] = (dx, dy) give it on the image area [x0, x1) x [y0, y1); precincts = [(w, h), ...],
        highest resolution first (opj_compress -c); prec / sgnd scalars or per-component lists."""
        self._sel()
        class Ext(C.Structure):
            _fields_ = [(n, C.c_int) for n in ("x0", "y0", "x1", "y1", "ncomp")] + \
                       [("dx", C.c_int * 8), ("dy", C.c_int * 8), ("prec", C.c_int * 8), ("sgnd", C.c_int * 8)] + \
                       [(n, C.c_int) for n in ("irreversible", "mct", "numres", "cblkw", "cblkh", "layers", "tile_w", "tile_h", "tx0", "ty0",
                                               "prog", "csty", "mode", "res_spec")] + \
                       [("prcw", C.c_int * 33), ("prch", C.c_int * 33), ("rsiz", C.c_int), ("max_cs_size", C.c_int), ("max_comp_size", C.c_int),
                        ("rates", C.c_float * 100), ("threads", C.c_int), ("roi_compno", C.c_int), ("roi_shift", C.c_int)]
        nc = len(comps)
        sub = sub or [(1, 1)] * nc
        precs = prec if isinstance(prec, (list, tuple)) else [prec] * nc
        sg = sgnd if isinstance(sgnd, (list, tuple)) else [int(bool(sgnd))] * nc
        e = Ext()
        e.x0, e.y0 = x0, y0
        e.x1 = x1 if x1 is not None else x0 + comps[0].shape[1] * sub[0][0]
        e.y1 = y1 if y1 is not None else y0 + comps[0].shape[0] * sub[0][1]
        e.ncomp = nc
        arrs = []
        for c in range(nc):
            e.dx[c], e.dy[c], e.prec[c], e.sgnd[c] = sub[c][0], sub[c][1], precs[c], sg[c]
            cw = -(-e.x1 // sub[c][0]) - -(-x0 // sub[c][0])
            ch = -(-e.y1 // sub[c][1]) - -(-y0 // sub[c][1])
            a = np.ascontiguousarray(comps[c], dtype=np.int32)
            assert a.shape == (ch, cw), (c, a.shape, (ch, cw))
            arrs.append(a)
        e.irreversible, e.mct, e.numres, e.cblkw, e.cblkh, e.layers = int(not reversible), int(mct), numres, cblk[0], cblk[1], layers
        e.tile_w, e.tile_h, e.tx0, e.ty0 = tile[0], tile[1], tile_origin[0], tile_origin[1]
        e.prog, e.csty, e.mode = prog, (2 if sop else 0) | (4 if eph else 0), mode
        if precincts:
            e.res_spec = len(precincts)
            for i, (pw, ph) in enumerate(precincts):
                e.prcw[i], e.prch[i] = pw, ph
        e.rsiz, e.max_cs_size, e.max_comp_size, e.threads = rsiz, max_cs_size, max_comp_size, threads
        e.roi_compno, e.roi_shift = (roi[0], roi[1]) if roi else (-1, 0)  # (component, shift): MAXSHIFT on the whole component
        e.rates[0] = -1.0
        if rates:
            e.layers = len(rates)
            for i, r in enumerate(rates):
                e.rates[i] = r
        ptrs = (C.POINTER(C.c_int32) * 8)(*[_i32p(a) for a in arrs])
        cap = sum(a.size for a in arrs) * 4 + (1 << 20)
        out = np.empty(cap, dtype=np.uint8)
        secs = C.c_double()
        self.L.opjr_encode_ext.restype = C.c_long
        self.L.opjr_encode_ext.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_double)]
        n = self.L.opjr_encode_ext(C.byref(e), ptrs, _u8p(out), cap, C.byref(secs))
        if n < 0:
            raise RuntimeError("openjpeg encode failed: " + self.L.opjr_last_error().decode())
        self.last_seconds = secs.value
        return out[:n].tobytes()

    def decode_comps(self, data: bytes, reduce: int = 0, threads: int = 0, apply_palette: bool = False):
        """Decode with per-component results: list of dict(data = 2-D int32 array, prec, sgnd, dx, dy, x0, y0).
        apply_palette: let libopenjp2 apply a JP2 palette itself (the reference asks it not to and hands the table to its host)."""
        self._sel()
        self.L.opjr_apply_palette(int(apply_palette))
        try:
            return self._decode_comps(data, reduce, threads)
        finally:
            self.L.opjr_apply_palette(0)

    def _decode_comps(self, data: bytes, reduce: int, threads: int):
        buf = np.frombuffer(data, dtype=np.uint8)
        cap = 1 << 26
        while True:
            out = np.empty(cap, dtype=np.int32)
            nc = C.c_int()
            dims = ((C.c_int * 8) * 16)()
            self.L.opjr_decode_comps.restype = C.c_int
            self.L.opjr_decode_comps.argtypes = [C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_int32), C.c_size_t, C.POINTER(C.c_int),
                                                 C.c_void_p, C.c_int, C.c_int]
            rc = self.L.opjr_decode_comps(_u8p(buf), len(data), _i32p(out), cap, C.byref(nc), dims, reduce, threads)
            if rc == -3 and cap < (1 << 31):
                cap <<= 2
                continue
            if rc != 0:
                raise RuntimeError("openjpeg decode failed: " + self.L.opjr_last_error().decode())
            break
        res, pos = [], 0
        for c in range(nc.value):
            w, h, prec, sg, dx, dy, cx0, cy0 = list(dims[c])
            res.append(dict(data=out[pos:pos + w * h].reshape(h, w).copy(), prec=prec, sgnd=sg, dx=dx, dy=dy, x0=cx0, y0=cy0))
            pos += w * h
        return res

    def set_progression(self, order: int):
        """Progression order of the following encodes (0 LRCP .. 4 CPRL = j2k::Order = OPJ_PROG_ORDER)."""
        self._sel()
        self.L.opjr_set_progression(order)

    @property
    def comment(self):
        return "Created by OpenJPEG version " + self.version

    def encode(self, planes: np.ndarray, params: Params, threads: int = 0) -> bytes:
        self._sel()
        planes = np.ascontiguousarray(planes, dtype=np.int32)
        assert planes.shape == (params.ncomp, params.height, params.width)
        assert params.tile_w == params.tile_h
        cap = planes.size * 4 + (1 << 20)
        out = np.empty(cap, dtype=np.uint8)
        secs = C.c_double()
        n = self.L.opjr_encode(_i32p(planes), params.width, params.height, params.ncomp, params.prec,
                               16 if params.prec > 8 else 8, int(not params.reversible), params.mct,
                               params.numres, 1 << params.cblkw_exp, 1 << params.cblkh_exp, params.layers,
                               params.tile_w, threads, _u8p(out), cap, C.byref(secs))
        if n < 0:
            raise RuntimeError("openjpeg encode failed: " + self.L.opjr_last_error().decode())
        self.last_seconds = secs.value
        return out[:n].tobytes()

    def encode_jp2(self, planes: np.ndarray, params: Params, color_space: int = -1, icc: bytes | None = None,
                   alpha_channel: int = -1, threads: int = 0) -> bytes:
        """JP2 file from libopenjp2's own JP2 writer (OPJ_CODEC_JP2): the reference's disabled branch
        (j2k_openjpeg_codec.cpp:613) with its colour-space mapping (:650-661); icc / alpha_channel fill
        opj_image_t::icc_profile_buf / comps[i].alpha.  color_space is an OPJ_COLOR_SPACE value."""
        self._sel()
        planes = np.ascontiguousarray(planes, dtype=np.int32)
        assert planes.shape == (params.ncomp, params.height, params.width)
        cap = planes.size * 4 + (1 << 20) + (len(icc) if icc else 0)
        out = np.empty(cap, dtype=np.uint8)
        secs = C.c_double()
        iccbuf = (C.c_uint8 * len(icc)).from_buffer_copy(icc) if icc else None
        n = self.L.opjr_encode_jp2(_i32p(planes), params.width, params.height, params.ncomp, params.prec,
                                   16 if params.prec > 8 else 8, int(not params.reversible), params.mct,
                                   params.numres, 1 << params.cblkw_exp, 1 << params.cblkh_exp, params.layers,
                                   params.tile_w, threads, color_space,
                                   C.cast(iccbuf, C.c_void_p) if icc else None, len(icc) if icc else 0, alpha_channel,
                                   _u8p(out), cap, C.byref(secs))
        if n < 0:
            raise RuntimeError("openjpeg JP2 encode failed: " + self.L.opjr_last_error().decode())
        return out[:n].tobytes()

    def encode_jp2_rates(self, planes: np.ndarray, params: Params, rates, color_space: int = -1, icc: bytes | None = None,
                         alpha_channel: int = -1) -> bytes:
        """JP2 file with rate control (encode_jp2 + encode_rates)."""
        self._sel()
        planes = np.ascontiguousarray(planes, dtype=np.int32)
        cap = planes.size * 4 + (1 << 20) + (len(icc) if icc else 0)
        out = np.empty(cap, dtype=np.uint8)
        secs = C.c_double()
        r = (C.c_float * len(rates))(*rates)
        iccbuf = (C.c_uint8 * len(icc)).from_buffer_copy(icc) if icc else None
        n = self.L.opjr_encode_jp2_rates(_i32p(planes), params.width, params.height, params.ncomp, params.prec,
                                         16 if params.prec > 8 else 8, int(not params.reversible), params.mct,
                                         params.numres, 1 << params.cblkw_exp, 1 << params.cblkh_exp, r, len(rates),
                                         params.tile_w, 0, color_space, C.cast(iccbuf, C.c_void_p) if icc else None,
                                         len(icc) if icc else 0, alpha_channel, _u8p(out), cap, C.byref(secs))
        if n < 0:
            raise RuntimeError("openjpeg JP2 encode failed: " + self.L.opjr_last_error().decode())
        return out[:n].tobytes()

    def encode_psnr(self, planes: np.ndarray, params: Params, psnr, threads: int = 0) -> bytes:
        """Fixed-quality encode: one PSNR target (dB) per layer (cp_fixed_quality, tcp_distoratio)."""
        self._sel()
        planes = np.ascontiguousarray(planes, dtype=np.int32)
        assert planes.shape == (params.ncomp, params.height, params.width)
        cap = planes.size * 4 + (1 << 20)
        out = np.empty(cap, dtype=np.uint8)
        secs = C.c_double()
        r = (C.c_float * len(psnr))(*psnr)
        n = self.L.opjr_encode_psnr(_i32p(planes), params.width, params.height, params.ncomp, params.prec,
                                    16 if params.prec > 8 else 8, int(not params.reversible), params.mct,
                                    params.numres, 1 << params.cblkw_exp, 1 << params.cblkh_exp, r, len(psnr),
                                    params.tile_w, threads, _u8p(out), cap, C.byref(secs))
        if n < 0:
            raise RuntimeError("openjpeg encode failed: " + self.L.opjr_last_error().decode())
        return out[:n].tobytes()

    def encode_rates(self, planes: np.ndarray, params: Params, rates, threads: int = 0) -> bytes:
        """Rate-controlled encode: one compression ratio per layer (tcp_rates, cp_disto_alloc); 0 = the rest."""
        self._sel()
        planes = np.ascontiguousarray(planes, dtype=np.int32)
        assert planes.shape == (params.ncomp, params.height, params.width)
        cap = planes.size * 4 + (1 << 20)
        out = np.empty(cap, dtype=np.uint8)
        secs = C.c_double()
        r = (C.c_float * len(rates))(*rates)
        n = self.L.opjr_encode_rates(_i32p(planes), params.width, params.height, params.ncomp, params.prec,
                                     16 if params.prec > 8 else 8, int(not params.reversible), params.mct,
                                     params.numres, 1 << params.cblkw_exp, 1 << params.cblkh_exp, r, len(rates),
                                     params.tile_w, threads, _u8p(out), cap, C.byref(secs))
        if n < 0:
            raise RuntimeError("openjpeg encode failed: " + self.L.opjr_last_error().decode())
        return out[:n].tobytes()

    def decode_ex(self, data: bytes, threads: int = 0):
        """Decode a raw codestream or a JP2 file; returns (planes, meta) with meta = dict(jp2, color_space,
        icc, alpha_mask) as libopenjp2 reports them after reading the boxes."""
        self._sel()
        buf = np.frombuffer(data, dtype=np.uint8)
        off = 0
        if data[:12] == b"\x00\x00\x00\x0cjP  \r\n\x87\n":
            off = data.index(b"jp2c") + 4
        w = int.from_bytes(data[off + 8:off + 12], "big")
        h = int.from_bytes(data[off + 12:off + 16], "big")
        nc = int.from_bytes(data[off + 40:off + 42], "big")
        out = np.empty((nc, h, w), dtype=np.int32)
        dims = (C.c_int * 4)()
        meta = (C.c_int * 4)()
        icc = np.zeros(1 << 20, dtype=np.uint8)
        rc = self.L.opjr_decode_ex(_u8p(buf), len(data), _i32p(out), out.size, dims, threads, meta, _u8p(icc), icc.size)
        if rc != 0:
            raise RuntimeError("openjpeg decode failed: " + self.L.opjr_last_error().decode())
        return out, {"jp2": bool(meta[0]), "color_space": meta[1], "icc": icc[:meta[2]].tobytes(), "alpha_mask": meta[3]}

    def decode_ref(self, data: bytes, reduce: int = 0, order: int = 0, threads: int = 0):
        """The reference's ReadFile call sequence (header first, then cp_reduce; order=1: OpenJPEG's documented
        order).  Returns (planes as the library left them, comps[0].factor)."""
        self._sel()
        buf = np.frombuffer(data, dtype=np.uint8)
        off = data.index(b"jp2c") + 4 if data[:12] == b"\x00\x00\x00\x0cjP  \r\n\x87\n" else 0
        w = int.from_bytes(data[off + 8:off + 12], "big")
        h = int.from_bytes(data[off + 12:off + 16], "big")
        nc = int.from_bytes(data[off + 40:off + 42], "big")
        out = np.zeros(nc * h * w, dtype=np.int32)
        dims = (C.c_int * 5)()
        self.L.opjr_decode_ref.restype = C.c_int
        self.L.opjr_decode_ref.argtypes = [C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_int32), C.c_size_t, C.POINTER(C.c_int),
                                           C.c_int, C.c_int, C.c_int]
        rc = self.L.opjr_decode_ref(_u8p(buf), len(data), _i32p(out), out.size, dims, threads, reduce, order)
        if rc != 0:
            raise RuntimeError(f"openjpeg decode failed ({rc}): " + self.L.opjr_last_error().decode())
        return out[:dims[2] * dims[1] * dims[0]].reshape(dims[2], dims[1], dims[0]).copy(), dims[4]

    def decode(self, cs: bytes, threads: int = 0) -> np.ndarray:
        self._sel()
        buf = np.frombuffer(cs, dtype=np.uint8)
        # SIZ: Xsiz,Ysiz at offset 8,12; Csiz at 40
        w = int.from_bytes(cs[8:12], "big")
        h = int.from_bytes(cs[12:16], "big")
        nc = int.from_bytes(cs[40:42], "big")
        out = np.empty((nc, h, w), dtype=np.int32)
        dims = (C.c_int * 4)()
        rc = self.L.opjr_decode(_u8p(buf), len(cs), _i32p(out), out.size, dims, threads)
        if rc != 0:
            raise RuntimeError("openjpeg decode failed: " + self.L.opjr_last_error().decode())
        return out


def strip_com(cs: bytes) -> bytes:
    """Remove COM (FF64) marker segments from the main header (version string differs by library)."""
    out = bytearray(cs[:2])
    i = 2
    while i < len(cs):
        m = cs[i:i + 2]
        if m == b"\xff\x90":  # SOT: main header over
            out += cs[i:]
            break
        ln = int.from_bytes(cs[i + 2:i + 4], "big")
        if m != b"\xff\x64":
            out += cs[i:i + 2 + ln]
        i += 2 + ln
    return bytes(out)

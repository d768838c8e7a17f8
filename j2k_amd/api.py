"""ctypes binding of the C ABI in include/j2k_hip.h (libj2k_hip.so).

This is harness plumbing for tests and bench.py; the product boundary is the C ABI itself and the
C++ `HipCodec` in j2k_amd/host/.  The binding fails loudly when the HIP library is missing -- there
is no CPU fallback.

torch is imported first (when present) so that this library binds to the same HIP runtime that
torch already loaded into the process.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

try:  # plumbing only: makes both libraries share one HIP runtime in-process
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

PKG = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.environ.get("J2K_HIP_LIB") or os.path.join(PKG, "libj2k_hip.so")  # (J2K_HIP_LIB: A/B runs of two builds on one box)


class J2kHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"j2k_hip error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("channels", C.c_uint32), ("depth", C.c_uint32), ("reversible", C.c_uint32),
                ("ycc", C.c_uint32), ("layers", C.c_uint32), ("tile_size", C.c_uint32),
                ("num_resolutions", C.c_uint32), ("cblk_w", C.c_uint32), ("cblk_h", C.c_uint32),
                ("progression", C.c_uint32), ("promote_ae16", C.c_uint32), ("comment", C.c_char_p),
                ("file_format", C.c_uint32), ("color_space", C.c_uint32), ("alpha", C.c_uint32),
                ("alpha_premultiplied", C.c_uint32), ("icc_profile", C.c_void_p), ("icc_profile_len", C.c_size_t),
                ("layer_rates", C.POINTER(C.c_float)), ("layer_psnr", C.POINTER(C.c_float)),
                ("pixel_aspect_num", C.c_uint32), ("pixel_aspect_den", C.c_uint32), ("dpi", C.c_float),
                ("num_precincts", C.c_uint32), ("precinct_w", C.c_uint32 * 33), ("precinct_h", C.c_uint32 * 33),
                ("dci_profile", C.c_uint32), ("max_cs_size", C.c_uint32), ("max_comp_size", C.c_uint32)]


class Plane(C.Structure):
    _fields_ = [("base", C.c_void_p), ("colbytes", C.c_ssize_t), ("rowbytes", C.c_ssize_t),
                ("sample_bits", C.c_uint32), ("depth", C.c_uint32)]


class FileInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("struct_size", "width", "height", "channels", "depth", "reversible", "ycc", "layers",
                                          "num_resolutions", "tile_width", "tile_height", "progression", "file_format",
                                          "color_space", "alpha", "alpha_premultiplied")] + \
               [("icc_profile_offset", C.c_size_t), ("icc_profile_len", C.c_size_t)] + \
               [(n, C.c_uint32 * 4) for n in ("sub_x", "sub_y", "comp_depth", "comp_signed")] + \
               [("lut_size", C.c_uint32), ("lut_channels", C.c_uint32), ("lut", (C.c_uint8 * 4) * 256), ("lut_column", C.c_uint8 * 4)]

    def as_dict(self):
        d = {n: (list(getattr(self, n)) if n in ("sub_x", "sub_y", "comp_depth", "comp_signed", "lut_column") else getattr(self, n))
             for n, _ in self._fields_ if n != "lut"}
        d["lut"] = [list(self.lut[i])[:self.lut_channels] for i in range(self.lut_size)]  # palette entries (JP2 pclr), [] without one
        return d


class OutPlane(C.Structure):
    _fields_ = [("base", C.c_void_p), ("colbytes", C.c_ssize_t), ("rowbytes", C.c_ssize_t), ("sample_bits", C.c_uint32),
                ("depth", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("ms_upload", "ms_frontend", "ms_dwt", "ms_t1", "ms_t2_host",
                                          "ms_assemble", "ms_download", "ms_total")] + \
               [("codestream_bytes", C.c_uint64), ("num_codeblocks", C.c_uint64), ("num_symbols", C.c_uint64),
                ("dwt_bytes", C.c_double), ("bands", C.c_uint32), ("reserved_", C.c_uint32), ("ms_after_upload", C.c_double),
                ("early_download_bytes", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


WRITE_FN = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t)

EXPORTS = ["j2k_hip_abi_version", "j2k_hip_create", "j2k_hip_destroy", "j2k_hip_last_error", "j2k_hip_encode",
           "j2k_hip_encode_begin", "j2k_hip_encode_begin_borrowed", "j2k_hip_encode_end", "j2k_hip_debug_tune", "j2k_hip_debug_get_tune", "j2k_hip_debug_fused_occupancy", "j2k_hip_debug_membw", "j2k_hip_debug_dwt_time", "j2k_hip_read_info", "j2k_hip_decode",
           "j2k_hip_decode_device", "j2k_hip_encode_tiles", "j2k_hip_device_count", "j2k_hip_encode_batch",
           "j2k_hip_encode_tiles_distributed", "j2k_hip_multi_last_error",
           "j2k_hip_encode_to_buffer", "j2k_hip_encode_device", "j2k_hip_encode_sequence_device", "j2k_hip_encode_tiles_device",
           "j2k_hip_main_header", "j2k_hip_file_header", "j2k_hip_stage_frontend", "j2k_hip_stage_dwt", "j2k_hip_stage_t1", "j2k_hip_stage_t1_passes",
           "j2k_hip_get_stats", "j2k_hip_get_dwt_level_ms", "j2k_hip_malloc", "j2k_hip_free",
           "j2k_hip_memcpy_h2d", "j2k_hip_memcpy_d2h", "j2k_hip_synchronize", "j2k_hip_debug_copy_sink", "j2k_hip_debug_count_sink"]

class CopySink(C.Structure):
    """include/j2k_hip.h: j2k_hip_copy_sink -- the `user` of j2k_hip_debug_copy_sink."""
    _fields_ = [("dst", C.c_void_p), ("capacity", C.c_size_t), ("pos", C.c_size_t)]


def native_sink(L, copying: bool = True):
    """The write function (a WRITE_FN) of a sink that lives in the library -- no Python in the write path.  Its `user`
    argument is the caller's: a CopySink (copying) or a c_size_t that receives the byte count (counting)."""
    fn = C.cast(L.j2k_hip_debug_copy_sink if copying else L.j2k_hip_debug_count_sink, WRITE_FN)
    return fn


_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBPATH):
        raise OSError(f"{LIBPATH} is missing: build it with `python __graft_entry__.py` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(LIBPATH)
    L.j2k_hip_last_error.restype = C.c_char_p
    L.j2k_hip_last_error.argtypes = [C.c_void_p]
    L.j2k_hip_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.j2k_hip_destroy.argtypes = [C.c_void_p]
    L.j2k_hip_debug_fused_occupancy.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.j2k_hip_destroy.restype = None
    L.j2k_hip_encode.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Plane), WRITE_FN, C.c_void_p]
    L.j2k_hip_encode_begin.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Plane)]
    L.j2k_hip_encode_end.argtypes = [C.c_void_p, WRITE_FN, C.c_void_p]
    L.j2k_hip_encode_begin_borrowed.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Plane)]
    L.j2k_hip_debug_tune.argtypes = [C.c_char_p, C.c_int]
    L.j2k_hip_debug_get_tune.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    L.j2k_hip_debug_membw.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.POINTER(C.c_double)]
    L.j2k_hip_encode_tiles.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Plane), C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t,
                                       C.POINTER(C.c_size_t)]
    L.j2k_hip_encode_batch.argtypes = [C.POINTER(C.c_int), C.c_uint32, C.c_uint32, C.POINTER(Params), C.POINTER(Plane), C.c_uint32,
                                       WRITE_FN, C.POINTER(C.c_void_p)]
    L.j2k_hip_encode_tiles_distributed.argtypes = [C.POINTER(C.c_int), C.c_uint32, C.POINTER(Params), C.POINTER(Plane), WRITE_FN, C.c_void_p]
    L.j2k_hip_multi_last_error.restype = C.c_char_p
    L.j2k_hip_read_info.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(FileInfo)]
    L.j2k_hip_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(OutPlane), C.c_uint32]
    L.j2k_hip_decode_device.argtypes = L.j2k_hip_decode.argtypes
    L.j2k_hip_debug_dwt_time.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
    L.j2k_hip_encode_to_buffer.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Plane), C.c_void_p, C.c_size_t,
                                           C.POINTER(C.c_size_t)]
    L.j2k_hip_encode_device.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Plane), C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t]
    L.j2k_hip_encode_sequence_device.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Plane), C.c_uint32,
                                                 C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.j2k_hip_encode_tiles_device.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Plane), C.c_uint32, C.c_uint32,
                                              C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t]
    L.j2k_hip_main_header.argtypes = [C.POINTER(Params), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                      C.POINTER(C.c_uint32)]
    L.j2k_hip_file_header.argtypes = [C.POINTER(Params), C.c_uint64, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.j2k_hip_stage_frontend.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Plane), C.c_void_p]
    L.j2k_hip_stage_dwt.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_double)]
    U32P = C.POINTER(C.c_uint32)
    L.j2k_hip_stage_t1.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, U32P, U32P, U32P, U32P, U32P,
                                   C.POINTER(C.c_float), U32P, U32P, U32P, C.POINTER(C.c_uint64), C.c_void_p, C.c_size_t]
    L.j2k_hip_stage_t1_passes.argtypes = L.j2k_hip_stage_t1.argtypes + [U32P, C.POINTER(C.c_int32)]
    L.j2k_hip_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.j2k_hip_get_dwt_level_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
    L.j2k_hip_malloc.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t]
    L.j2k_hip_free.argtypes = [C.c_void_p, C.c_void_p]
    L.j2k_hip_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.j2k_hip_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.j2k_hip_synchronize.argtypes = [C.c_void_p]
    _lib = L
    return L


def tune(key: str, value: int):
    """Process-wide tuning knob of the library (j2k_hip_debug_tune); never changes an output byte."""
    L = load_library()
    if L.j2k_hip_debug_tune(key.encode(), int(value)) != 0:
        raise KeyError(key)


def get_tune(key: str) -> int:
    L = load_library()
    v = C.c_int()
    if L.j2k_hip_debug_get_tune(key.encode(), C.byref(v)) != 0:
        raise KeyError(key)
    return v.value


def read_info(data: bytes) -> dict:
    """Header of a raw codestream or JP2 file (j2k_hip_read_info; no device needed)."""
    L = load_library()
    fi = FileInfo()
    fi.struct_size = C.sizeof(FileInfo)
    buf = np.frombuffer(data, dtype=np.uint8)
    rc = L.j2k_hip_read_info(buf.ctypes.data, len(data), C.byref(fi))
    if rc != 0:
        raise J2kHipError(rc, L.j2k_hip_last_error(None).decode())
    return fi.as_dict()


def make_params(width, height, channels, depth, reversible=True, ycc=False, layers=1, tile_size=0,
                num_resolutions=6, cblk=(64, 64), promote=False, comment="", jp2=False, color_space=0,
                alpha_channel=-1, alpha_premultiplied=False, icc=None, rates=None, psnr=None, progression=0,
                pixel_aspect=None, dpi=0.0, precincts=None, dci_profile=0, max_cs_size=0, max_comp_size=0):
    """comment: None -> library default COM, "" -> no COM segment.  jp2/color_space/alpha_channel/icc describe
    the JP2 file wrapper (color_space in OPJ_COLOR_SPACE numbering: 1 sRGB, 2 grey, 3 sYCC)."""
    p = Params()
    p.struct_size = C.sizeof(Params)
    p.width, p.height, p.channels, p.depth = width, height, channels, depth
    p.reversible, p.ycc, p.layers, p.tile_size = int(reversible), int(ycc), layers, tile_size
    p.num_resolutions, p.cblk_w, p.cblk_h = num_resolutions, cblk[0], cblk[1]
    p.progression, p.promote_ae16 = progression, int(promote)
    p.comment = comment.encode() if comment is not None else None
    p.file_format, p.color_space = int(jp2), color_space
    p.alpha, p.alpha_premultiplied = alpha_channel + 1, int(alpha_premultiplied)
    if pixel_aspect:
        p.pixel_aspect_num, p.pixel_aspect_den = pixel_aspect
    p.dpi = dpi
    p.dci_profile, p.max_cs_size, p.max_comp_size = dci_profile, max_cs_size, max_comp_size  # 3 / 4: OpenJPEG's cinema 2K / 4K profile
    if precincts:  # [(w, h), ...] highest resolution first (OpenJPEG's -c / res_spec semantics)
        p.num_precincts = len(precincts)
        for i, (pw, ph) in enumerate(precincts):
            p.precinct_w[i], p.precinct_h[i] = pw, ph
    if rates is not None:  # one compression ratio per layer (OpenJPEG tcp_rates); sets the layer count
        p.layers = len(rates)
        p._rates_keepalive = (C.c_float * len(rates))(*rates)
        p.layer_rates = C.cast(p._rates_keepalive, C.POINTER(C.c_float))
    if psnr is not None:  # one PSNR target (dB) per layer (OpenJPEG tcp_distoratio / cp_fixed_quality)
        p.layers = len(psnr)
        p._psnr_keepalive = (C.c_float * len(psnr))(*psnr)
        p.layer_psnr = C.cast(p._psnr_keepalive, C.POINTER(C.c_float))
    if icc:
        p._icc_keepalive = C.create_string_buffer(bytes(icc), len(icc))  # borrowed by the C side for each call
        p.icc_profile, p.icc_profile_len = C.cast(p._icc_keepalive, C.c_void_p), len(icc)
    return p


def main_header(params: Params) -> bytes:
    """SOC..QCD[,COM] of the codestream (what rank 0 of a tile-sharded job prepends)."""
    L = load_library()
    n, nt = C.c_size_t(), C.c_uint32()
    buf = C.create_string_buffer(70000)
    rc = L.j2k_hip_main_header(C.byref(params), buf, len(buf), C.byref(n), C.byref(nt))
    if rc != 0:
        raise J2kHipError(rc, L.j2k_hip_last_error(None).decode())
    return buf.raw[:n.value]


def file_header(params: Params, codestream_len: int) -> bytes:
    """Bytes that precede the codestream in the output file (JP2 boxes; empty for raw J2K)."""
    L = load_library()
    n = C.c_size_t()
    buf = C.create_string_buffer(4096 + int(params.icc_profile_len))
    rc = L.j2k_hip_file_header(C.byref(params), codestream_len, buf, len(buf), C.byref(n))
    if rc != 0:
        raise J2kHipError(rc, L.j2k_hip_last_error(None).decode())
    return buf.raw[:n.value]


def planes_from_layout(base_addr: int, layout: dict, channels: int, depth_bits: int | None = None):
    """Channel views (R,G,B[,A] = codec channels 0..) over an AE ARGB frame (see synth.ae_frame)."""
    sb = layout["sample_bytes"]
    offs = layout["channel_offsets"]  # A,R,G,B
    order = [offs[1], offs[2], offs[3], offs[0]]
    arr = (Plane * channels)()
    for c in range(channels):
        arr[c].base = base_addr + order[c]
        arr[c].colbytes = layout["colbytes"]
        arr[c].rowbytes = layout["rowbytes"]
        arr[c].sample_bits = 8 * sb
        arr[c].depth = depth_bits if depth_bits is not None else 8 * sb
    return arr


class Encoder:
    """Thin RAII wrapper over a j2k_hip_encoder handle."""

    def __init__(self, device: int = 0):
        self.L = load_library()
        self.h = C.c_void_p()
        rc = self.L.j2k_hip_create(C.byref(self.h), device)
        if rc != 0:
            raise J2kHipError(rc, self.L.j2k_hip_last_error(None).decode())

    def close(self):
        if getattr(self, "h", None) and self.h:
            self.L.j2k_hip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise J2kHipError(rc, self.L.j2k_hip_last_error(self.h).decode())

    # -- memory -----------------------------------------------------------------------------------
    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._check(self.L.j2k_hip_malloc(self.h, C.byref(p), nbytes))
        return p.value

    def free(self, dptr: int):
        self._check(self.L.j2k_hip_free(self.h, dptr))

    def h2d(self, dptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self._check(self.L.j2k_hip_memcpy_h2d(self.h, dptr, arr.ctypes.data, arr.nbytes))

    def d2h(self, dptr: int, nbytes: int) -> np.ndarray:
        out = np.empty(nbytes, dtype=np.uint8)
        self._check(self.L.j2k_hip_memcpy_d2h(self.h, out.ctypes.data, dptr, nbytes))
        return out

    def upload(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr)
        d = self.malloc(max(arr.nbytes, 16))
        self.h2d(d, arr)
        return d

    def synchronize(self):
        self._check(self.L.j2k_hip_synchronize(self.h))

    # -- encode -----------------------------------------------------------------------------------
    def encode_host(self, frame: np.ndarray, layout: dict, params: Params, via_sink: bool = False) -> bytes:
        planes = planes_from_layout(frame.ctypes.data, layout, params.channels)
        return self._encode_planes_host(planes, frame.nbytes, params, via_sink)

    def encode_planar_host(self, planes_arr: np.ndarray, params: Params, via_sink: bool = False) -> bytes:
        """planes_arr: (channels, h, w) unsigned samples -> planar host buffers of 8- or 16-bit samples."""
        dt = np.uint16 if params.depth > 8 else np.uint8
        buf = np.ascontiguousarray(planes_arr.astype(dt))
        nc, h, w = buf.shape
        arr = (Plane * nc)()
        for c in range(nc):
            arr[c].base = buf.ctypes.data + c * h * w * buf.itemsize
            arr[c].colbytes, arr[c].rowbytes = buf.itemsize, w * buf.itemsize
            arr[c].sample_bits, arr[c].depth = 8 * buf.itemsize, params.depth  # samples already hold `depth` bits
        return self._encode_planes_host(arr, buf.nbytes, params, via_sink)

    def _encode_planes_host(self, planes, in_bytes: int, params: Params, via_sink: bool) -> bytes:
        if via_sink:
            chunks = []

            @WRITE_FN
            def sink(user, buf, n):
                chunks.append(C.string_at(buf, n))
                return n
            self._check(self.L.j2k_hip_encode(self.h, C.byref(params), planes, sink, None))
            return b"".join(chunks)
        cap = in_bytes * 2 + (1 << 20) + int(params.icc_profile_len)
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t()
        self._check(self.L.j2k_hip_encode_to_buffer(self.h, C.byref(params), planes, out.ctypes.data, cap, C.byref(n)))
        return out[:n.value].tobytes()

    def encode_begin_host(self, frame: np.ndarray, layout: dict, params: Params):
        """First half of j2k_hip_encode (j2k_hip_encode_begin): returns once the frame has left `frame`."""
        planes = planes_from_layout(frame.ctypes.data, layout, params.channels)
        self._check(self.L.j2k_hip_encode_begin(self.h, C.byref(params), planes))

    def encode_begin_borrowed(self, frame: np.ndarray, layout: dict, params: Params):
        """j2k_hip_encode_begin_borrowed: returns at once; `frame` and `params` must stay as they are until encode_end()."""
        planes = planes_from_layout(frame.ctypes.data, layout, params.channels)
        self._borrowed = (frame, params, planes)  # (kept alive on the caller's behalf)
        self._check(self.L.j2k_hip_encode_begin_borrowed(self.h, C.byref(params), planes))

    def encode_end(self) -> bytes:
        """Second half (j2k_hip_encode_end): the finished file through the sink callback."""
        chunks = []

        @WRITE_FN
        def sink(user, buf, n):
            chunks.append(C.string_at(buf, n))
            return n
        self._check(self.L.j2k_hip_encode_end(self.h, sink, None))
        return b"".join(chunks)

    def encode_device(self, d_frame: int, layout: dict, params: Params, download: bool = True):
        """Returns (device_ptr, length, bytes or None)."""
        planes = planes_from_layout(d_frame, layout, params.channels)
        dptr, n = C.c_void_p(), C.c_size_t()
        self._check(self.L.j2k_hip_encode_device(self.h, C.byref(params), planes, C.byref(dptr), C.byref(n), None, 0))
        data = self.d2h(dptr.value, n.value).tobytes() if download else None
        return dptr.value, n.value, data

    def encode_sequence_device(self, d_frames: list, layout: dict, params: Params, download: bool = True):
        """Frames of one sequence (device pointers, same layout) in one call -> [(device_ptr, length, bytes or None)]."""
        nf, nc = len(d_frames), params.channels
        arr = (Plane * (nf * nc))()
        for f, d in enumerate(d_frames):
            one = planes_from_layout(d, layout, nc)
            for c in range(nc):
                arr[f * nc + c] = one[c]
        ptrs, lens = (C.c_void_p * nf)(), (C.c_size_t * nf)()
        self._check(self.L.j2k_hip_encode_sequence_device(self.h, C.byref(params), arr, nf, ptrs, lens))
        return [(ptrs[f], lens[f], self.d2h(ptrs[f], lens[f]).tobytes() if download else None) for f in range(nf)]

    def encode_tiles_device(self, d_frame: int, layout: dict, params: Params, tile_first: int, tile_count: int):
        planes = planes_from_layout(d_frame, layout, params.channels)
        dptr, n = C.c_void_p(), C.c_size_t()
        self._check(self.L.j2k_hip_encode_tiles_device(self.h, C.byref(params), planes, tile_first, tile_count,
                                                       C.byref(dptr), C.byref(n), None, 0))
        return self.d2h(dptr.value, n.value).tobytes()

    # -- decode -----------------------------------------------------------------------------------
    def decode_planar(self, data: bytes, subsample: int = 1, sample_bits: int | None = None, depth: int | None = None,
                      channels: int | None = None, out: np.ndarray | None = None) -> np.ndarray:
        """Decode into planar host buffers: (channels, ceil(h / subsample), ceil(w / subsample)) of uint8 / uint16.
        depth = Channel.depth of the destination (default: the file's precision in the smallest fitting sample type).
        out: a buffer of that shape to decode into (a host that decodes frame after frame keeps its buffers)."""
        i = read_info(data)
        nc = channels or i["channels"]
        red = max(subsample, 1).bit_length() - 1
        w, h = -(-i["width"] >> red), -(-i["height"] >> red)
        bits = sample_bits or (8 if i["depth"] <= 8 else 16)
        if out is None:
            out = np.zeros((nc, h, w), dtype=np.uint8 if bits == 8 else np.uint16)
        assert out.shape == (nc, h, w) and out.itemsize * 8 == bits and out.flags.c_contiguous
        arr = (OutPlane * nc)()
        for c in range(nc):
            arr[c].base = out.ctypes.data + c * h * w * out.itemsize
            arr[c].colbytes, arr[c].rowbytes = out.itemsize, w * out.itemsize
            arr[c].sample_bits, arr[c].depth = bits, depth or min(i["depth"], bits)
            arr[c].width, arr[c].height = w, h
        buf = np.frombuffer(data, dtype=np.uint8)
        self._check(self.L.j2k_hip_decode(self.h, buf.ctypes.data, len(data), subsample, arr, nc))
        return out

    def decode_channels(self, data: bytes, chans: list, depth: int | None = None, subsample: int = 1):
        """Decode into one 2-D numpy view per codec channel, wherever each lies and whatever its strides (padded rows,
        samples of interleaved pixels, bottom-up rows): the general form of the C ABI's destination."""
        arr = (OutPlane * len(chans))()
        for c, a in enumerate(chans):
            assert a.ndim == 2 and a.dtype in (np.uint8, np.uint16)
            arr[c].base = a.ctypes.data
            arr[c].colbytes, arr[c].rowbytes = a.strides[1], a.strides[0]
            arr[c].sample_bits, arr[c].depth = 8 * a.itemsize, depth or 8 * a.itemsize
            arr[c].width, arr[c].height = a.shape[1], a.shape[0]
        buf = np.frombuffer(data, dtype=np.uint8)
        self._check(self.L.j2k_hip_decode(self.h, buf.ctypes.data, len(data), subsample, arr, len(chans)))

    def decode_ae(self, data: bytes, frame: np.ndarray, layout: dict, width: int, height: int, channels: int, depth: int | None = None,
                  subsample: int = 1, device: bool = False):
        """Decode into an After Effects ARGB frame (see synth.ae_frame): codec channels R,G,B[,A] go to their samples,
        every other byte of `frame` must stay as it is.  device=True: through a device copy of the frame."""
        sb = layout["sample_bytes"]
        offs = layout["channel_offsets"]
        order = [offs[1], offs[2], offs[3], offs[0]]
        base = frame.ctypes.data
        d = None
        if device:
            d = self.upload(frame)
            base = d
        arr = (OutPlane * channels)()
        for c in range(channels):
            arr[c].base = base + order[c]
            arr[c].colbytes, arr[c].rowbytes = layout["colbytes"], layout["rowbytes"]
            arr[c].sample_bits, arr[c].depth = 8 * sb, depth or 8 * sb
            arr[c].width, arr[c].height = width, height
        buf = np.frombuffer(data, dtype=np.uint8)
        try:
            if device:
                self._check(self.L.j2k_hip_decode_device(self.h, buf.ctypes.data, len(data), subsample, arr, channels))
                frame[:] = self.d2h(d, frame.nbytes)
            else:
                self._check(self.L.j2k_hip_decode(self.h, buf.ctypes.data, len(data), subsample, arr, channels))
        finally:
            if d:
                self.free(d)
        return frame

    def stats(self) -> dict:
        s = Stats()
        self._check(self.L.j2k_hip_get_stats(self.h, C.byref(s)))
        return s.as_dict()

    def dwt_time(self, first: int, count: int, repeat: int = 20) -> float:
        """Mean device time (ms) of the DWT launches of levels [first, first+count) of the last encode, replayed back to back."""
        ms = C.c_double()
        self._check(self.L.j2k_hip_debug_dwt_time(self.h, first, count, repeat, C.byref(ms)))
        return ms.value

    def dwt_level_ms(self):
        buf = (C.c_double * 40)()
        n = self.L.j2k_hip_get_dwt_level_ms(self.h, buf, 40)
        return list(buf[:n])

    # -- stages -----------------------------------------------------------------------------------
    def stage_frontend(self, frame: np.ndarray, layout: dict, params: Params) -> np.ndarray:
        d_in = self.upload(frame)
        n = params.channels * params.width * params.height
        d_out = self.malloc(4 * n)
        try:
            planes = planes_from_layout(d_in, layout, params.channels)
            self._check(self.L.j2k_hip_stage_frontend(self.h, C.byref(params), planes, d_out))
            raw = self.d2h(d_out, 4 * n)
        finally:
            self.free(d_in)
            self.free(d_out)
        dt = np.int32 if params.reversible else np.float32
        return raw.view(dt).reshape(params.channels, params.height, params.width)

    def stage_dwt(self, planes: np.ndarray, levels: int, reversible: bool, x0=0, y0=0, repeat=1):
        """planes: (n, h, w) int32 / float32. Returns (result, ms per run)."""
        dt = np.int32 if reversible else np.float32
        planes = np.ascontiguousarray(planes, dtype=dt)
        n, h, w = planes.shape
        d_in = self.upload(planes)
        d_out = self.malloc(planes.nbytes)
        ms = C.c_double()
        try:
            self._check(self.L.j2k_hip_stage_dwt(self.h, int(reversible), w, h, n, levels, x0, y0, d_in, d_out, repeat,
                                                 C.byref(ms)))
            raw = self.d2h(d_out, planes.nbytes)
        finally:
            self.free(d_in)
            self.free(d_out)
        return raw.view(dt).reshape(n, h, w), ms.value

    def stage_t1(self, coef: np.ndarray, rects, orients, stepsizes, reversible: bool, want_passes: bool = False):
        """coef: (H, W) int32/float32 plane; rects: list of (x, y, w, h). Returns list of dicts."""
        dt = np.int32 if reversible else np.float32
        coef = np.ascontiguousarray(coef, dtype=dt)
        H, W = coef.shape
        nb = len(rects)
        U = lambda v: (C.c_uint32 * nb)(*v)
        bx, by, bw, bh = (U([r[i] for r in rects]) for i in range(4))
        ori = U(orients)
        ss = (C.c_float * nb)(*stepsizes)
        numbps, npasses, length = U([0] * nb), U([0] * nb), U([0] * nb)
        offs = (C.c_uint64 * nb)()
        cap = sum(r[2] * r[3] for r in rects) * 8 + 4096
        data = np.empty(cap, dtype=np.uint8)
        d = self.upload(coef)
        MP = 96
        rates = (C.c_uint32 * (nb * MP))()
        dist = (C.c_int32 * (nb * MP))()
        try:
            if want_passes:
                self._check(self.L.j2k_hip_stage_t1_passes(self.h, int(reversible), d, W, nb, bx, by, bw, bh, ori, ss, numbps,
                                                           npasses, length, offs, data.ctypes.data, cap, rates, dist))
            else:
                self._check(self.L.j2k_hip_stage_t1(self.h, int(reversible), d, W, nb, bx, by, bw, bh, ori, ss, numbps, npasses,
                                                    length, offs, data.ctypes.data, cap))
        finally:
            self.free(d)
        out = [dict(numbps=numbps[i], npasses=npasses[i], length=length[i],
                    data=data[offs[i]:offs[i] + length[i]].tobytes()) for i in range(nb)]
        if want_passes:
            for i, o in enumerate(out):
                o["rates"] = list(rates[i * MP:i * MP + o["npasses"]])
                o["nmsedec"] = list(dist[i * MP:i * MP + o["npasses"]])
        return out

"""Seeded synthetic inputs and the After Effects frame layout (row A0).

The generator is an LCG iterated once per sample, component-major then row-major:
    s <- 1664525*s + 1013904223 (mod 2^32)
    v(c,x,y) = ((((3x + 5y + 977c) << (prec-8)) / 4) + (s >> (32 - nbits))) & (2^prec - 1)
distribution "A": nbits = prec-4 (high entropy); distribution "B": nbits = 2 (smooth).

The AE layout mirrors WorldToBuffer (reference: src/aftereffects/j2k.cpp:324-362): one interleaved
A,R,G,B buffer, 8-bit -> ARGB32, deeper -> ARGB64 with the value left-justified to 16 bits so that
CopyBuffer's `>> (16 - prec)` (reference: src/common/j2k_codec.cpp:355-371) recovers it.
"""
from __future__ import annotations

import numpy as np

_A = np.uint64(1664525)
_C = np.uint64(1013904223)
_M = np.uint64(0xFFFFFFFF)
_CHUNK = 1 << 20
_tab = None


def _jump_table():
    """(A_j, C_j) with s_{n+j} = A_j*s_n + C_j (mod 2^32) for j = 1.._CHUNK."""
    global _tab
    if _tab is None:
        a = np.array([_A], dtype=np.uint64)
        c = np.array([_C], dtype=np.uint64)
        while a.size < _CHUNK:
            al, cl = a[-1], c[-1]
            a2 = (a * al) & _M
            c2 = (a * cl + c) & _M
            a = np.concatenate([a, a2])
            c = np.concatenate([c, c2])
        _tab = (a[:_CHUNK], c[:_CHUNK])
    return _tab


def lcg_stream(seed: int, n: int) -> np.ndarray:
    """First n LCG outputs (state after each step) as uint32."""
    a, c = _jump_table()
    out = np.empty(n, dtype=np.uint32)
    s = np.uint64(seed & 0xFFFFFFFF)
    for off in range(0, n, _CHUNK):
        m = min(_CHUNK, n - off)
        blk = (a[:m] * s + c[:m]) & _M
        out[off:off + m] = blk.astype(np.uint32)
        s = blk[m - 1]
    return out


def planes(width: int, height: int, ncomp: int, prec: int, seed: int, dist: str = "A") -> np.ndarray:
    """(ncomp, height, width) int32 unsigned sample values."""
    nbits = (prec - 4) if dist == "A" else 2
    s = lcg_stream(seed, ncomp * height * width).reshape(ncomp, height, width)
    x = np.arange(width, dtype=np.int64)[None, None, :]
    y = np.arange(height, dtype=np.int64)[None, :, None]
    c = np.arange(ncomp, dtype=np.int64)[:, None, None]
    grad = ((3 * x + 5 * y + 977 * c) << (prec - 8)) // 4
    noise = (s >> np.uint32(32 - nbits)).astype(np.int64) if nbits > 0 else 0
    return ((grad + noise) & ((1 << prec) - 1)).astype(np.int32)


def ae_frame(pl: np.ndarray, prec: int, row_pad_bytes: int = 0) -> tuple[np.ndarray, dict]:
    """Pack 1..4 planes (R,G,B[,A] order; a single plane goes to R,G,B) into the AE ARGB layout.

    Returns (buffer uint8 1-D, layout) where layout = dict(pixel_size, colbytes, rowbytes,
    channel_offsets = byte offset of A,R,G,B inside a pixel, sample_bytes).
    """
    ncomp, h, w = pl.shape
    sb = 1 if prec <= 8 else 2
    dt = np.uint8 if sb == 1 else np.uint16
    full = (1 << (8 * sb)) - 1
    shift = 8 * sb - prec
    rowbytes = 4 * sb * w + row_pad_bytes
    buf = np.zeros(h * rowbytes, dtype=np.uint8)
    view = np.lib.stride_tricks.as_strided(buf.view(dt) if sb == 2 else buf, shape=(h, w, 4),
                                           strides=(rowbytes, 4 * sb, sb), writeable=True)
    view[:, :, 0] = full if ncomp < 4 else (pl[3].astype(np.int64) << shift).astype(dt)
    for i in range(3):
        src = pl[i] if ncomp >= 3 else pl[0]
        view[:, :, 1 + i] = (src.astype(np.int64) << shift).astype(dt)
    return buf, dict(sample_bytes=sb, colbytes=4 * sb, rowbytes=rowbytes,
                     channel_offsets=(0, sb, 2 * sb, 3 * sb))

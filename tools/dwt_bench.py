#!/usr/bin/env python3
"""Micro-benchmark of the DWT kernel alone (stage entry point): one level of 3 planes SxS."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from j2k_amd import api
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rev = len(sys.argv) > 3 and sys.argv[3] == "53"
enc = api.Encoder(0)
rng = np.random.default_rng(1)
a = (rng.standard_normal((3, S, S)) * 1000).astype(np.int32 if rev else np.float32)
_, ms = enc.stage_dwt(a, levels, rev, repeat=1)
_, ms = enc.stage_dwt(a, levels, rev, repeat=20)
nbytes = 8.0 * 3 * S * S * sum(0.25 ** l for l in range(levels))
print(f"PAIRS={os.environ.get('J2K_DWT_PAIRS','2')} PF={os.environ.get('J2K_DWT_PF','1')} PPC={os.environ.get('J2K_DWT_PPC','auto')} "
      f"S={S} levels={levels} {'5/3' if rev else '9/7'}: {ms*1e3:.1f} us  {nbytes/ms/1e6:.0f} GB/s")
if os.environ.get("J2K_COPY_CAL"):
    import torch, time
    x = torch.empty(3 * S * S, dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
    y.copy_(x); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): y.copy_(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"torch D2D copy {x.numel()*8/dt/1e9:.0f} GB/s (read+write)")
if os.environ.get("J2K_MEMBW"):
    import ctypes as C
    g = C.c_double()
    enc.L.j2k_hip_debug_membw.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.POINTER(C.c_double)]
    for mode, rows in [(0, 0), (2, 0), (3, 0), (1, 64), (1, 256)]:
        enc.L.j2k_hip_debug_membw(enc.h, 3 * S if mode != 1 else S, S, rows, mode, 20, C.byref(g))
        print(f"membw mode={mode} rows={rows}: {g.value:.0f} GB/s (read+write)")

#!/usr/bin/env python3
"""End-to-end encode rate of the other BASELINE.json configs on one MI355X (frame resident in HBM ->
codestream in HBM), for DESIGN.md.  Usage: python tools/bench_configs.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np
from j2k_amd import api, synth

CONFIGS = [
    ("C1 512x512 grey8 5/3", 512, 512, 1, 8, dict(reversible=True, ycc=False, num_resolutions=6), 12345),
    ("C2 4096x4096 RGB8 9/7 5lvl", 4096, 4096, 3, 8, dict(reversible=False, ycc=True, num_resolutions=6), 12345),
    ("C3 8192x8192 RGB16 9/7 6lvl", 8192, 8192, 3, 16, dict(reversible=False, ycc=True, num_resolutions=7), 23456),
    ("C3 8192x8192 RGB16 9/7 5lvl", 8192, 8192, 3, 16, dict(reversible=False, ycc=True, num_resolutions=6), 23456),
    ("C4/8: 16384x2048 RGB16 5/3 tiles 2048 (8 of 64 tiles)", 16384, 2048, 3, 16, dict(reversible=True, ycc=True, num_resolutions=6, tile_size=2048), 34567),
    ("C5 frame 4096x2160 RGB10 9/7", 4096, 2160, 3, 10, dict(reversible=False, ycc=True, num_resolutions=6), 45678),
    # what After Effects really sends for 16-bit projects: an ARGB64 world of 15+1-bit samples, alpha as the 4th channel, Promote on load
    ("AE RGB16+promote 8192x8192 9/7 5lvl", 8192, 8192, 3, 16, dict(reversible=False, ycc=True, num_resolutions=6, promote=True), 23456),
    ("AE RGBA16+promote 8192x8192 9/7 5lvl", 8192, 8192, 4, 16, dict(reversible=False, ycc=True, num_resolutions=6, promote=True), 23456),
    ("AE RGBA16 8192x8192 9/7 5lvl (no promote)", 8192, 8192, 4, 16, dict(reversible=False, ycc=True, num_resolutions=6), 23456),
    ("ref-literal 4096x4096 RGB8 5/3 no MCT tile 1024 12 layers", 4096, 4096, 3, 8, dict(reversible=True, ycc=False, num_resolutions=6, tile_size=1024, layers=12), 12345),
]
import threading
NFL = int(os.environ.get("NFL", "3"))
encs = [api.Encoder(0) for _ in range(NFL)]
enc = encs[0]
ONLY = os.environ.get("ONLY")
for name, w, h, nc, prec, kw, seed in CONFIGS:
    if ONLY and not name.startswith(ONLY): continue
    SEQ = int(os.environ.get("SEQ", "0"))
    pl = synth.planes(w, h, nc, prec, seed)
    if kw.get("promote"):
        pl >>= 1  # 15+1-bit samples: Promote gives the 16-bit values back (up to the lost bit)
    frame, lay = synth.ae_frame(pl, prec)
    del pl
    d = enc.upload(frame)
    p = api.make_params(w, h, nc, prec, comment="", **kw)
    for e in encs:
        e.encode_device(d, lay, p, download=False)
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        _, ln, _ = enc.encode_device(d, lay, p, download=False)
    dt = (time.perf_counter() - t0) / n
    st = enc.stats()
    lv = enc.dwt_level_ms()
    # the same frames through NFL handles on NFL host threads (frames in flight)
    per = 8
    def worker(e):
        for _ in range(per):
            e.encode_device(d, lay, p, download=False)
    ths = [threading.Thread(target=worker, args=(e,)) for e in encs]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dtf = (time.perf_counter() - t0) / (per * NFL)
    print(f"{name}: one at a time {w*h/dt/1e6:8.1f} Mpixel/s {dt*1e3:7.2f} ms (frontend={st['ms_frontend']:.3f} dwt={st['ms_dwt']:.3f} level1={lv[0] if lv else 0:.3f} t1={st['ms_t1']:.2f} t2host={st['ms_t2_host']:.2f}); "
          f"{NFL} in flight {w*h/dtf/1e6:8.1f} Mpixel/s = {nc*w*h/dtf/1e6:8.1f} Msample/s {dtf*1e3:7.2f} ms; bytes={ln}", flush=True)
    if SEQ:  # image sequence: SEQ frames per call (j2k_hip_encode_sequence_device), on NFL handles
        def seq_worker(e):
            for _ in range(4):
                e.encode_sequence_device([d] * SEQ, lay, p, download=False)
        for e in encs: e.encode_sequence_device([d] * SEQ, lay, p, download=False)
        ths = [threading.Thread(target=seq_worker, args=(e,)) for e in encs]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        dts = (time.perf_counter() - t0) / (4 * NFL * SEQ)
        print(f"    sequences of {SEQ} frames per call, {NFL} calls in flight: {w*h/dts/1e6:8.1f} Mpixel/s {dts*1e3:7.2f} ms per frame", flush=True)
    enc.free(d)

#!/usr/bin/env python3
"""Decode throughput with several frames in flight: N host threads, a handle each (what a host that reads frames from
several threads does), the same file decoded into per-thread planar buffers.  Prints frames/s and Mpixel/s for both Tier-1
decoders (t1dec_lanes 0 = a wave per code-block, 2 = a lane per code-block) and the default choice (1).
usage: tools/decode_inflight.py [CASE ...]   (C5, C2, C3, C4tile)"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")  # a hardware queue per stream of the handles in flight (as bench.py; must precede HIP's start)
import numpy as np  # noqa: E402

from j2k_amd import api, synth  # noqa: E402

CASES = {"C2": (4096, 4096, 3, 8, False), "C5": (4096, 2160, 3, 10, False), "C4tile": (2048, 2048, 3, 16, True), "C3": (8192, 8192, 3, 16, False), "DCI4K": (4096, 2160, 3, 12, False)}
THREADS = {"C3": (1, 2, 3, 4), "C2": (1, 2, 4, 8), "C5": (1, 2, 4, 8, 12), "C4tile": (1, 4, 8), "DCI4K": (1, 2, 4, 8)}
enc = api.Encoder(0)
for name in (sys.argv[1:] or ["C5", "C2", "C3"]):
    w, h, nc, prec, rev = CASES[name]
    pl = synth.planes(w, h, nc, prec, 7)
    frame, lay = synth.ae_frame(pl, prec)
    p = api.make_params(w, h, nc, prec, reversible=rev, ycc=True, comment="")
    if name == "DCI4K":
        p = api.make_params(w, h, nc, prec, num_resolutions=7, dci_profile=4, comment="")
    cs = enc.encode_host(frame, lay, p)
    ref = enc.decode_planar(cs)
    del frame
    for lanes in [int(v) for v in os.environ.get("LANES_ORDER", "0,2,1").split(",")]:
        api.tune("t1dec_lanes", lanes)
        for nt in THREADS[name]:
            per = max(2, 24 // nt) if name != "C3" else 3
            encs = [api.Encoder(0) for _ in range(nt)]
            outs = [None] * nt

            def work(i, count):
                for _ in range(count):
                    outs[i] = encs[i].decode_planar(cs, out=outs[i])

            ths = [threading.Thread(target=work, args=(i, 1)) for i in range(nt)]  # warm-up: arenas, destination pages
            [t.start() for t in ths]
            [t.join() for t in ths]
            t0 = time.perf_counter()
            ths = [threading.Thread(target=work, args=(i, per)) for i in range(nt)]
            [t.start() for t in ths]
            [t.join() for t in ths]
            dt = time.perf_counter() - t0
            ok = all(np.array_equal(o, ref) for o in outs)
            print(f"{name} t1dec_lanes={lanes} threads={nt}: {nt * per / dt:7.1f} frames/s = {nt * per * w * h / dt / 1e6:7.0f} Mpixel/s, {dt / (nt * per) * 1e3:6.1f} ms per frame"
                  f"{'' if ok else '  ** OUTPUT DIFFERS **'}", flush=True)
            for e in encs:
                e.close()
    api.tune("t1dec_lanes", 1)
    del pl, ref
enc.close()

#!/usr/bin/env python3
"""Rate-controlled encode of the metric frame with frames in flight: Mpixel/s for a number of handles / host threads.
usage: rate_inflight.py [frames_in_flight ...]     (J2K_ALLOC_THREADS etc. from the environment)"""
import os, sys, time, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from j2k_amd import api, synth
S = 8192
pl = synth.planes(S, S, 3, 16, 23456); frame, lay = synth.ae_frame(pl, 16); del pl
up = api.Encoder(0)
d = up.upload(frame)
p = api.make_params(S, S, 3, 16, reversible=False, ycc=True, num_resolutions=6, comment="", rates=[20.0])
for nfl in [int(x) for x in sys.argv[1:]] or [3, 6]:
    encs = [api.Encoder(0) for _ in range(nfl)]
    for e in encs: e.encode_device(d, lay, p, download=False)
    per = 6
    def worker(e):
        for _ in range(per): e.encode_device(d, lay, p, download=False)
    ths = [threading.Thread(target=worker, args=(e,)) for e in encs]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = (time.perf_counter() - t0) / (per * nfl)
    print(f"alloc_threads={os.environ.get('J2K_ALLOC_THREADS', 'default')} {nfl} frames in flight: {dt*1e3:.1f} ms/frame = {S*S/dt/1e6:.0f} Mpixel/s", flush=True)
    for e in encs: e.close()

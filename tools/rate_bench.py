#!/usr/bin/env python3
"""Rate-controlled encode of the metric frame (8192^2 RGB16 9/7): where the time goes.
usage: rate_bench.py [size] [ratio ...]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")  # one hardware queue per stream of the handles in flight (before HIP starts)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from j2k_amd import api, synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rates = [float(x) for x in sys.argv[2:]] or [20.0]
pl = synth.planes(S, S, 3, 16, 23456); frame, lay = synth.ae_frame(pl, 16); del pl
enc = api.Encoder(0)
d = enc.upload(frame)
for r in (None, rates):
    p = api.make_params(S, S, 3, 16, reversible=False, ycc=True, num_resolutions=6, comment="", rates=r)
    enc.encode_device(d, lay, p, download=False)
    t0 = time.perf_counter(); n = 3
    for _ in range(n): _, ln, _ = enc.encode_device(d, lay, p, download=False)
    dt = (time.perf_counter() - t0) / n
    st = enc.stats()
    print(f"rates={r}: {dt*1e3:.1f} ms/frame, {ln} bytes (ratio {S*S*6/ln:.2f}); dwt {st['ms_dwt']:.2f} t1 {st['ms_t1']:.1f} t2_host {st['ms_t2_host']:.1f} assemble {st['ms_assemble']:.1f}", flush=True)

# the same with frames in flight (NFL handles on NFL host threads): the host-side layer allocation of one
# frame overlaps the GPU work and the allocations of the others
import threading
for nfl in (3, 6):
    encs = [api.Encoder(0) for _ in range(nfl)]
    p = api.make_params(S, S, 3, 16, reversible=False, ycc=True, num_resolutions=6, comment="", rates=rates)
    for e in encs: e.encode_device(d, lay, p, download=False)
    per = 3
    def worker(e):
        for _ in range(per): e.encode_device(d, lay, p, download=False)
    ths = [threading.Thread(target=worker, args=(e,)) for e in encs]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = (time.perf_counter() - t0) / (per * nfl)
    print(f"rates={rates}, {nfl} frames in flight: {dt*1e3:.1f} ms/frame = {S*S/dt/1e6:.0f} Mpixel/s", flush=True)
    for e in encs: e.close()

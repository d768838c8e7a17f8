#!/usr/bin/env python3
"""Rate-controlled encode of the metric frame (8192^2 RGB16 9/7): where the time goes.
usage: rate_bench.py [size] [ratio ...]"""
import os, sys, time
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

#!/usr/bin/env python3
"""PCIe-inclusive rate of the plug-in path: host frame in (pageable), codestream out to a host buffer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "5")
from j2k_amd import api, synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
pl = synth.planes(S, S, 3, 16, 23456); frame, lay = synth.ae_frame(pl, 16); del pl
enc = api.Encoder(0)
p = api.make_params(S, S, 3, 16, reversible=False, ycc=True, num_resolutions=6, comment="")
enc.encode_host(frame, lay, p)
for via in (False, True):
    t0 = time.perf_counter(); n = 3
    for _ in range(n): cs = enc.encode_host(frame, lay, p, via_sink=via)
    dt = (time.perf_counter() - t0) / n
    st = enc.stats()
    print(f"host->host {'sink' if via else 'buffer'}: {S*S/dt/1e6:.1f} Mpixel/s, {dt*1e3:.1f} ms/frame (upload {st['ms_upload']:.1f} ms, download {st['ms_download']:.1f} ms, total {st['ms_total']:.1f})", flush=True)

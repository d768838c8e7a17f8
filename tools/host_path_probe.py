#!/usr/bin/env python3
"""PCIe-inclusive rate of the plug-in path: host frame in (pageable), codestream out to a host buffer.
usage: host_path_probe.py [size] [threads] [frames-per-thread]"""
import os, sys, time, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")  # one hardware queue per stream of the handles in flight (before HIP starts)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from j2k_amd import api, synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 1
NF = int(sys.argv[3]) if len(sys.argv) > 3 else 4
pl = synth.planes(S, S, 3, 16, 23456); frame, lay = synth.ae_frame(pl, 16); del pl
p = api.make_params(S, S, 3, 16, reversible=False, ycc=True, num_resolutions=6, comment="")
encs = [api.Encoder(0) for _ in range(NT)]
frames = [frame.copy() for _ in range(NT)]
for e, f in zip(encs, frames): e.encode_host(f, lay, p)
for via in (False, True):
    def worker(k):
        for _ in range(NF): encs[k].encode_host(frames[k], lay, p, via_sink=via)
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(NT)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = (time.perf_counter() - t0) / (NT * NF)
    st = encs[0].stats()
    print(f"host->host {'sink' if via else 'buffer'} x{NT} threads: {S*S/dt/1e6:.1f} Mpixel/s, {dt*1e3:.1f} ms/frame "
          f"(per frame: upload {st['ms_upload']:.1f} ms, download {st['ms_download']:.1f} ms, total {st['ms_total']:.1f})", flush=True)

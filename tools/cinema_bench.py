#!/usr/bin/env python3
"""The coding the CINEMA method maps to (HipCodec::HonourSettings, aftereffects/j2k.cpp:817-830): 4096 x 2160 RGB12 in the 4K
digital cinema profile (dci_profile = 4: Rsiz 4, 9/7, 7 resolutions, 32 x 32 blocks, CPRL, precincts 128 / 256, six tile-parts,
TLM, POC), cut to the DCI frame budget (250 Mbit/s at 24 fps = 1,302,083 bytes, 1,041,666 per component); PROFILE=0: the same
coding style as plain rate control (no profile flag, one tile-part).  Frames per second one at a time and with handles in
flight.   usage: cinema_bench.py [frames_in_flight ...]"""
import os, sys, time, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from j2k_amd import api, synth
W, H, PREC, BUDGET = 4096, 2160, 12, 1302083
pl = synth.planes(W, H, 3, PREC, 45678); frame, lay = synth.ae_frame(pl, PREC); del pl
ratio = W * H * 3 * PREC / 8.0 / BUDGET
if os.environ.get("PROFILE", "4") == "4":
    p = api.make_params(W, H, 3, PREC, num_resolutions=7, dci_profile=4, comment="")
else:
    p = api.make_params(W, H, 3, PREC, reversible=False, ycc=True, num_resolutions=7, cblk=(32, 32), progression=4, comment="", rates=[ratio],
                        precincts=[(256, 256)] * 6 + [(128, 128)])
enc = api.Encoder(0)
d = enc.upload(frame)
enc.encode_device(d, lay, p, download=False)
n = 6
t0 = time.perf_counter()
for _ in range(n): _, ln, _ = enc.encode_device(d, lay, p, download=False)
dt = (time.perf_counter() - t0) / n
st = enc.stats()
print(f"DCI 4K frame, one at a time: {dt*1e3:.1f} ms/frame = {1/dt:.0f} frames/s, {ln} bytes (budget {BUDGET}); "
      f"dwt {st['ms_dwt']:.2f} t1 {st['ms_t1']:.1f} t2_host {st['ms_t2_host']:.1f}", flush=True)
for nfl in [int(x) for x in sys.argv[1:]] or [3, 6]:
    encs = [api.Encoder(0) for _ in range(nfl)]
    for e in encs: e.encode_device(d, lay, p, download=False)
    per = 12
    def worker(e):
        for _ in range(per): e.encode_device(d, lay, p, download=False)
    ths = [threading.Thread(target=worker, args=(e,)) for e in encs]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = (time.perf_counter() - t0) / (per * nfl)
    print(f"DCI 4K frames, {nfl} in flight: {dt*1e3:.2f} ms/frame = {1/dt:.0f} frames/s = {W*H/dt/1e6:.0f} Mpixel/s", flush=True)
    for e in encs: e.close()

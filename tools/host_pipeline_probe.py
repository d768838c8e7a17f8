#!/usr/bin/env python3
"""Where one host thread's time goes when it pipelines three handles through j2k_hip_encode_begin / _end
(pageable 8192^2 RGB16 frame in, counting sink out): mean duration of each call and the handle's stage times."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
from j2k_amd import api, synth  # noqa: E402

S = 8192
pl = synth.planes(S, S, 3, 16, 23456)
frame, lay = synth.ae_frame(pl, 16)
del pl
params = api.make_params(S, S, 3, 16, reversible=False, ycc=True, num_resolutions=6, comment="")
planes = api.planes_from_layout(frame.ctypes.data, lay, 3)
encs = [api.Encoder(0) for _ in range(3)]
pos = [0]


@api.WRITE_FN
def sink(user, p, n):
    pos[0] += n
    return n


for e in encs:
    e._check(e.L.j2k_hip_encode(e.h, C.byref(params), planes, sink, None))
total = 18
tb, te = [], []
t0 = time.perf_counter()
for i in range(total + 2):
    if i >= 2:
        k = (i - 2) % 3
        t = time.perf_counter()
        encs[k]._check(encs[k].L.j2k_hip_encode_end(encs[k].h, sink, None))
        te.append((time.perf_counter() - t) * 1e3)
        if i == total:
            print("stats of one frame:", {a: round(b, 2) for a, b in encs[k].stats().items() if a.startswith("ms_")})
    if i < total:
        k = i % 3
        t = time.perf_counter()
        encs[k]._check(encs[k].L.j2k_hip_encode_begin(encs[k].h, C.byref(params), planes))
        tb.append((time.perf_counter() - t) * 1e3)
dt = time.perf_counter() - t0
print(f"{S * S * total / dt / 1e6:.0f} Mpixel/s, {dt / total * 1e3:.2f} ms per frame | begin {sum(tb[3:]) / len(tb[3:]):.2f} ms, end {sum(te[3:]) / len(te[3:]):.2f} ms")
for e in encs:
    e.close()

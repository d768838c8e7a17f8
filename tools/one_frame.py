#!/usr/bin/env python3
"""One metric frame resident in HBM, encoded N times on one handle (nothing else on the device): the smallest command to put under
rocprofv3 --pmc / --kernel-trace when only the kernels of a frame are wanted.  usage: one_frame.py [size] [frames] [levels]"""
import hashlib
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from j2k_amd import api, synth

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
LV = int(sys.argv[3]) if len(sys.argv) > 3 else 5
pl = synth.planes(S, S, 3, 16, 23456)
frame, lay = synth.ae_frame(pl, 16)
del pl
p = api.make_params(S, S, 3, 16, reversible=False, ycc=True, num_resolutions=LV + 1, comment="")
e = api.Encoder(0)
d = e.upload(frame)
for _ in range(N):
    dptr, n, _ = e.encode_device(d, lay, p, download=False)
print("sha256", hashlib.sha256(e.d2h(dptr, n)).hexdigest(), "stats", {k: round(v, 3) if isinstance(v, float) else v for k, v in e.stats().items()})
e.free(d)
e.close()

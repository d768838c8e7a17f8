import os, sys, time, threading
sys.path.insert(0, "/root/repo")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np
from j2k_amd import api, synth
S=8192
pl = synth.planes(S, S, 3, 16, 23456); frame, lay = synth.ae_frame(pl, 16); del pl
encs = [api.Encoder(0) for _ in range(2)]
d = encs[0].upload(frame)
p = api.make_params(S, S, 3, 16, reversible=False, ycc=True, num_resolutions=6, comment="")
for e in encs: e.encode_device(d, lay, p, download=False)
lv = []
def worker(k):
    for i in range(4):
        encs[k].encode_device(d, lay, p, download=False)
        lv.append((k, [round(x,3) for x in encs[k].dwt_level_ms()], round(encs[k].stats()["ms_t1"],2)))
ths = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
t0=time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
print("Mpix/s", 8*S*S/(time.perf_counter()-t0)/1e6)
for x in lv: print(x)

#!/usr/bin/env python3
"""Device memory after a handle has come and gone, six times over, with rate-controlled and plain encodes of a 4096^2 RGB16 frame:
free memory must stand still from the second round on (the runtime keeps its own pools after the first).  usage: tools/leak_probe.py"""
import os, sys, json, hashlib
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from j2k_amd import api, synth
golden = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
g = golden["rh1_4096_rgb16_97_r20"]
pl = synth.planes(g["width"], g["height"], g["ncomp"], g["prec"], g["seed"], g["dist"])
frame, lay = synth.ae_frame(pl, g["prec"]); del pl
kw = g["params"]
torch.cuda.init()
for mode in ("rates", "plain"):
    p = api.make_params(g["width"], g["height"], g["ncomp"], g["prec"], reversible=False, ycc=True, num_resolutions=kw.get("numres", 6), comment=g["comment"], rates=g["rates"] if mode == "rates" else None)
    for it in range(6):
        e = api.Encoder(0)
        for _ in range(3):
            cs = e.encode_host(frame, lay, p)
        e.close()
        torch.cuda.synchronize()
        print(mode, it, "free MiB", torch.cuda.mem_get_info(0)[0] >> 20, flush=True)

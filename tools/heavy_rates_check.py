"""One-off check (not in the test tiers: the oracle needs ~10 s per case at this size): rate control and fixed quality on a\nframe large enough for the two coder groups and, with J2K_MQ_HEAVY=30000, for many blocks on the scalar coder."""
import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from j2k_amd import api, synth
from oracle import oracle as O
w, h = 4096, 3072
pl = synth.planes(w, h, 3, 16, 77, "A")
orc = O.Oracle()
enc = api.Encoder(0)
frame, lay = synth.ae_frame(pl, 16)
# (9216 blocks: the layer allocation's per-block work runs on the device, rate.hip; J2K_RATE_DEV=-1 keeps it on the host)
for mode, vals in (("rates", [24.0, 6.0]), ("rates", [100.0, 30.0, 10.0, 4.0]), ("rates", [8.0]), ("rates", [400.0, 150.0]), ("psnr", [50.0, 70.0])):
    t0 = time.time()
    p = O.make_params(w, h, 3, 16, reversible=False, mct=True, numres=6, layers=len(vals))
    ref = orc.encode_rates(pl, p, vals, comment="x") if mode == "rates" else orc.encode_psnr(pl, p, vals, comment="x")
    t1 = time.time()
    hp = api.make_params(w, h, 3, 16, reversible=False, ycc=True, num_resolutions=6, comment="x", **{mode: vals})
    got = enc.encode_host(frame, lay, hp)
    st = enc.stats()
    print(mode, vals, len(ref), len(got), got == ref, f"oracle {t1-t0:.1f}s gpu {st['ms_total']:.1f} ms blocks {st['num_codeblocks']}", flush=True)

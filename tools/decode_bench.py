#!/usr/bin/env python3
"""Decode path timing on one GPU: encode a BASELINE-shaped frame on the GPU, then decode it several times
(host file bytes -> planar host channels, and into a device buffer) and print the stage times of j2k_hip_stats."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from j2k_amd import api, synth  # noqa: E402

CASES = {"C2": (4096, 4096, 3, 8, False), "C5": (4096, 2160, 3, 10, False), "C4tile": (2048, 2048, 3, 16, True), "C3": (8192, 8192, 3, 16, False), "DCI4K": (4096, 2160, 3, 12, False)}
enc = api.Encoder(0)
for name in (sys.argv[1:] or ["C2", "C5", "C4tile", "C3"]):
    w, h, nc, prec, rev = CASES[name]
    pl = synth.planes(w, h, nc, prec, 7)
    frame, lay = synth.ae_frame(pl, prec)
    p = api.make_params(w, h, nc, prec, reversible=rev, ycc=True, comment="")
    if name == "DCI4K":  # a frame of the 4K digital cinema profile: 32 x 32 blocks, 1.3 MB
        p = api.make_params(w, h, nc, prec, num_resolutions=7, dci_profile=4, comment="")
    cs = enc.encode_host(frame, lay, p)
    for sub in (1, 2, 4):
        ts = []
        out = None
        for it in range(4):  # (the destination is kept from call to call, like a host's frame buffer: no fresh pages in the timing)
            t0 = time.perf_counter()
            out = enc.decode_planar(cs, subsample=sub, out=out)
            ts.append((time.perf_counter() - t0) * 1e3)
        st = enc.stats()
        if sub == 1 and name == "C3":  # what the plug-in's ReadFile does: R, G, B into the host's ARGB64 world (A kept: masked merge on the host)
            world = np.zeros_like(frame)
            ta = []
            for it in range(3):
                t0 = time.perf_counter()
                enc.decode_ae(cs, world, lay, w, h, 3)
                ta.append((time.perf_counter() - t0) * 1e3)
            print(f"{name} into the ARGB64 world: call {min(ta[1:]):.1f} ms", flush=True)
        print(f"{name} {w}x{h} prec {prec} {'5/3' if rev else '9/7'} subsample {sub}: call {min(ts[1:]):.1f} ms = {w * h / min(ts[1:]) / 1e3:.0f} Mpix/s "
              f"(full-size pixels) | host tier-2 {st['ms_t2_host']:.1f}, upload {st['ms_upload']:.2f}, gather+t1 {st['ms_t1']:.1f}, idwt {st['ms_dwt']:.2f}, "
              f"output {st['ms_frontend']:.2f} ms | {len(cs) / 1e6:.1f} MB, {st['num_codeblocks']} blocks", flush=True)
    if rev:
        assert np.array_equal(enc.decode_planar(cs), pl)
    del pl, frame
enc.close()

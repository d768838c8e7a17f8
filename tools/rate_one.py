import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from j2k_amd import api, synth
S = 8192
pl = synth.planes(S, S, 3, 16, 23456); frame, lay = synth.ae_frame(pl, 16); del pl
enc = api.Encoder(0)
d = enc.upload(frame)
p = api.make_params(S, S, 3, 16, reversible=False, ycc=True, num_resolutions=6, comment="", rates=[20.0])
for i in range(3):
    sys.stderr.write(f"--- frame {i}\n"); sys.stderr.flush()
    t0 = time.perf_counter()
    enc.encode_device(d, lay, p, download=False)
    st = enc.stats()
    sys.stderr.write(f"frame {i}: {(time.perf_counter()-t0)*1e3:.1f} ms t1 {st['ms_t1']:.1f} t2_host {st['ms_t2_host']:.1f}\n")

#!/usr/bin/env python3
"""The synchronous host call (j2k_hip_encode = what HipCodec::WriteFile makes) with the frame uploaded in row bands
(knob `bands`: -1 = one piece, n = n bands): ms per frame with a counting and with a copying native sink, and the
call's own account (j2k_hip_stats).  usage: band_probe.py [size] [frames] [bands,bands,...] [threads]"""
import ctypes as C
import hashlib
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from j2k_amd import api, synth

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 6
BANDS = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [-1, 1, 2, 4, 8]
NT = int(sys.argv[4]) if len(sys.argv) > 4 else 1
pl = synth.planes(S, S, 3, 16, 23456)
frame, lay = synth.ae_frame(pl, 16)
del pl
p = api.make_params(S, S, 3, 16, reversible=False, ycc=True, num_resolutions=6, comment="")
L = api.load_library()
encs = [api.Encoder(0) for _ in range(NT)]
frames = [frame] + [frame.copy() for _ in range(NT - 1)]
cap = frame.nbytes
ref = None


SINKS = {}


def sink_of(e, copying):  # (the sink's memory is the host's and not fresh on every frame: allocated and touched outside the timing)
    key = (id(e), copying)
    if key not in SINKS:
        if copying:
            buf = np.empty(cap, dtype=np.uint8)
            buf[::4096] = 0
            st = api.CopySink(buf.ctypes.data, cap, 0)
            SINKS[key] = (buf, st, C.cast(C.pointer(st), C.c_void_p))
        else:
            cnt = C.c_size_t(0)
            SINKS[key] = (None, cnt, C.cast(C.pointer(cnt), C.c_void_p))
    return SINKS[key]


def run(e, fr, copying, count, keep=None):
    planes = api.planes_from_layout(fr.ctypes.data, lay, 3)
    fn = api.native_sink(L, copying)
    buf, st, user = sink_of(e, copying)
    for _ in range(count):
        if copying:
            st.pos = 0
        e._check(L.j2k_hip_encode(e.h, C.byref(p), planes, fn, user))
    if copying and keep is not None:
        keep.append(hashlib.sha256(buf[:st.pos].tobytes()).hexdigest())


for b in BANDS:
    api.tune("bands", b)
    hashes = []
    for e, fr in zip(encs, frames):
        run(e, fr, False, 1)
        run(e, fr, True, 1, hashes)  # warm-up + the bytes
    ref = ref or hashes[0]
    line = f"bands {b:2d}: {'ok' if set(hashes) == {ref} else 'HASH MISMATCH'}"
    for copying in (False, True):
        ths = [threading.Thread(target=run, args=(e, fr, copying, NF)) for e, fr in zip(encs, frames)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = (time.perf_counter() - t0) / (NF * NT)
        st = encs[0].stats()
        line += (f" | {'copying' if copying else 'counting'} sink {dt * 1e3:6.2f} ms/frame = {S * S / dt / 1e6:6.0f} Mpixel/s"
                 f" (upload {st['ms_upload']:.2f}, after upload {st['ms_after_upload']:.2f}, gpu span {st['ms_t1']:.2f}, t2 {st['ms_t2_host']:.2f},"
                 f" tail {st['ms_assemble']:.2f}, dl wait {st['ms_download']:.2f}, early {st['early_download_bytes'] / 1e6:.0f} MB, bands {st['bands']})")
    print(line, flush=True)
api.tune("bands", 0)
for e in encs:
    e.close()

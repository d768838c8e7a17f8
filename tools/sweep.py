#!/usr/bin/env python3
"""One-process sweeps of the library's tuning knobs on the metric frame (8192^2 RGB16 9/7, 5 levels).

    python tools/sweep.py membw          copy ceilings of this box (j2k_hip_debug_membw, several shapes and sizes)
    python tools/sweep.py dwt            the DWT launches alone (one frame at a time, per-level hipEvents) under knob variants
    python tools/sweep.py live           3 frames in flight: Mpixel/s and the live DWT figures under knob variants (CU masks ...)
    python tools/sweep.py upload         host frame upload: pageable copy vs pinned staging pieces

Knobs go through j2k_hip_debug_tune (no output byte depends on a knob; every variant's codestream hash is checked
against the first one).  Output: one line per variant, also appended to gpurun_out/sweep_<section>.txt.
"""
import ctypes as C
import hashlib
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

import numpy as np  # noqa: E402

from j2k_amd import api, synth  # noqa: E402

S = int(os.environ.get("SWEEP_SIZE", "8192"))
PREC, LEVELS = 16, 5
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)


def emit(section, line):
    print(line, flush=True)
    with open(os.path.join(OUT, f"sweep_{section}.txt"), "a") as f:
        f.write(line + "\n")


def frame_on_device(enc):
    pl = synth.planes(S, S, 3, PREC, 23456)
    frame, lay = synth.ae_frame(pl, PREC)
    del pl
    d = enc.upload(frame)
    return frame, lay, d


def params():
    return api.make_params(S, S, 3, PREC, reversible=False, ycc=True, num_resolutions=LEVELS + 1, comment="")


def membw():
    enc = api.Encoder(0)
    g = C.c_double()
    for (w, h) in [(8192, 8192), (3 * 8192, 8192), (4 * 8192, 16384)]:
        for mode, rows in [(0, 0), (0, 1024), (0, 4096), (0, 8192), (2, 0), (2, 2048), (2, 8192), (3, 0), (3, 8192), (4, 0), (1, 16), (1, 48), (1, 256), (5, 16), (5, 48), (5, 256)]:
            ww = w if mode not in (1, 5) else min(w, 8192)
            enc._check(enc.L.j2k_hip_debug_membw(enc.h, ww, h, rows, mode, 20, C.byref(g)))
            emit("membw", f"membw {ww}x{h} floats ({ww * h * 4 / 2**20:.0f} MiB each way) mode={mode} grid/rows={rows}: {g.value:.0f} GB/s read+write")
    try:
        import torch
        for n in (1 << 26, 1 << 28, 1 << 29):
            x = torch.empty(n, dtype=torch.float32, device="cuda")
            y = torch.empty_like(x)
            y.copy_(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                y.copy_(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            emit("membw", f"torch D2D copy {n * 4 / 2**20:.0f} MiB: {n * 8 / dt / 1e9:.0f} GB/s read+write")
            del x, y
    except Exception as ex:  # noqa: BLE001
        emit("membw", f"torch copy skipped: {ex}")
    enc.close()


def set_knobs(kn):
    for k, v in kn.items():
        api.tune(k, v)


DEFAULTS = dict(dwt_xcd=1, fused_ppc=0, fused_wpb=0, dwt_ppc=0, dwt_min_waves=2048, dwt_pairs=2, dwt_depth=1, coder_cus=0,
                level_events=0, dwt_nt=0, dwt_ntl=0, mq_wait_us=1500, mq_yield=2, mq_prio=1, dwt_ahead=0, groups=2, heavy_min=0)


def dwt():
    """DWT launches replayed back to back (j2k_hip_debug_dwt_time): no events between launches, 40 replays each."""
    enc = api.Encoder(0)
    frame, lay, d = frame_on_device(enc)
    del frame
    p = params()
    ref = None
    variants = [dict(), dict(fused_ppc=12), dict(fused_ppc=20), dict(dwt_min_waves=3072)]
    if os.environ.get("SWEEP_PPC"):  # chunk lengths on other frame sizes (SWEEP_SIZE): is 16 row pairs per chunk right beyond the 8K frame?
        variants = [dict(fused_ppc=v) for v in (0, 8, 12, 16, 24, 32, 64)] + [dict(dwt_min_waves=v) for v in (1024, 2048, 4096)]
        if "," in os.environ["SWEEP_PPC"]:  # an explicit list of chunk lengths
            variants = [dict(fused_ppc=int(v)) for v in os.environ["SWEEP_PPC"].split(",")]
    l1 = 8.0 * 3 * S * S
    tot = l1 * sum(0.25 ** k for k in range(LEVELS))
    for kn in variants:
        set_knobs({**DEFAULTS, **kn})
        dptr, n, _ = enc.encode_device(d, lay, p, download=False)
        h = hashlib.sha256(enc.d2h(dptr, n)).hexdigest()
        ref = ref or h
        t1 = min(enc.dwt_time(0, 1, 40) for _ in range(3))
        tr = min(enc.dwt_time(1, LEVELS - 1, 40) for _ in range(3))
        ta = min(enc.dwt_time(0, LEVELS, 40) for _ in range(3))
        per = [min(enc.dwt_time(l, 1, 40) for _ in range(2)) for l in range(1, LEVELS)]
        emit("dwt", f"{'ok ' if h == ref else 'HASH MISMATCH '}{kn}: L1 {t1 * 1e3:.1f} us = {l1 / t1 / 1e6:.0f} GB/s frac {l1 / t1 / 1e6 / 8000:.3f} | "
                    f"L2..5 {tr * 1e3:.1f} us ({[round(x * 1e3, 1) for x in per]} each alone) | all {ta * 1e3:.1f} us = {tot / ta / 1e6:.0f} GB/s frac {tot / ta / 1e6 / 8000:.3f}")
    set_knobs(DEFAULTS)
    enc.free(d)
    enc.close()


def live():
    boot = api.Encoder(0)
    frame, lay, d = frame_on_device(boot)
    del frame
    p = params()
    planes = api.planes_from_layout(d, lay, 3)
    ref = None
    variants = [dict(), dict(mq_yield=0), dict()]
    if os.environ.get("SWEEP_LIVE"):  # an explicit list of knob settings (JSON), e.g. '[{}, {"mq_yield": 0}]'
        import json
        variants = json.loads(os.environ["SWEEP_LIVE"])
    for kn in variants:
        kn = dict(kn)
        want_nfl = kn.pop("inflight", None)
        for nfl in ((want_nfl,) if want_nfl else (3, 4) if kn.get("coder_cus") else (3,)):
            set_knobs({**DEFAULTS, **kn})
            encs = [api.Encoder(0) for _ in range(nfl)]
            outs = [(C.c_void_p(), C.c_size_t()) for _ in range(nfl)]
            stats = []

            def run(count):
                stats.clear()

                def worker(k):
                    e = encs[k]
                    for i in range(k, count, nfl):
                        e._check(e.L.j2k_hip_encode_device(e.h, C.byref(p), planes, C.byref(outs[k][0]), C.byref(outs[k][1]), None, 0))
                        if k == 0:
                            stats.append((e.stats(), e.dwt_level_ms()))
                ths = [threading.Thread(target=worker, args=(k,)) for k in range(nfl)]
                for t in ths:
                    t.start()
                for t in ths:
                    t.join()
            run(2 * nfl)
            t0 = time.perf_counter()
            steps = 36
            run(steps)
            boot.synchronize()
            dt = time.perf_counter() - t0
            hs = {hashlib.sha256(encs[k].d2h(outs[k][0].value, outs[k][1].value)).hexdigest() for k in range(nfl)}
            ref = ref or next(iter(hs))
            lv = np.array([x[1] for x in stats])
            l1 = 8.0 * 3 * S * S
            tot = l1 * sum(0.25 ** k for k in range(LEVELS))
            m = lv.mean(axis=0)
            emit("live", f"{'ok ' if hs == {ref} else 'HASH MISMATCH '}{kn} inflight={nfl}: {S * S * steps / dt / 1e6:.0f} Mpix/s {dt / steps * 1e3:.2f} ms/frame | "
                         f"live L1 {m[0]:.3f} ms frac {l1 / m[0] / 1e6 / 8000:.3f} | live phase {m.sum():.3f} ms frac {tot / m.sum() / 1e6 / 8000:.3f} | "
                         f"t1 {np.mean([x[0]['ms_t1'] for x in stats]):.2f} ms, latency {np.mean([x[0]['ms_total'] for x in stats]):.1f} ms")
            for e in encs:
                e.close()
    set_knobs(DEFAULTS)
    boot.free(d)
    boot.close()


def upload():
    enc = api.Encoder(0)
    pl = synth.planes(S, S, 3, PREC, 23456)
    frame, lay = synth.ae_frame(pl, PREC)
    del pl
    p = params()
    sink_buf = (C.c_uint8 * frame.nbytes)()
    pos = [0]

    @api.WRITE_FN
    def sink(user, ptr, n):
        C.memmove(C.addressof(sink_buf) + pos[0], ptr, n)
        pos[0] += n
        return n
    planes = api.planes_from_layout(frame.ctypes.data, lay, 3)
    for kn in (dict(staging=0), dict(staging=1, stage_kb=4096), dict(staging=1, stage_kb=16384), dict(staging=1, stage_kb=65536)):
        set_knobs(kn)
        ms = []
        for it in range(5):
            pos[0] = 0
            t0 = time.perf_counter()
            enc._check(enc.L.j2k_hip_encode(enc.h, C.byref(p), planes, sink, None))
            ms.append((time.perf_counter() - t0) * 1e3)
            st = enc.stats()
        emit("upload", f"{kn}: call {np.median(ms[1:]):.1f} ms, upload {st['ms_upload']:.2f} ms ({frame.nbytes / st['ms_upload'] / 1e6:.1f} GB/s), "
                       f"download wait {st['ms_download']:.2f} ms")
    set_knobs(dict(staging=0))
    enc.close()


if __name__ == "__main__":
    for sec in sys.argv[1:] or ["membw", "dwt", "live"]:
        emit(sec, f"# {sec} {time.strftime('%Y-%m-%d %H:%M:%S')} size {S}")
        dict(membw=membw, dwt=dwt, live=live, upload=upload)[sec]()

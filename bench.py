#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X JPEG 2000 encode path.

Metric (BASELINE.json): end-to-end Mpixels/s of the encode hot path on 8192x8192 16-bit RGB,
9/7 irreversible + ICT, 5 DWT levels, 64x64 code-blocks -- plus the achieved HBM GB/s of the 9/7 DWT
kernel against the chip's roofline.

A "step" = one pass of the whole hot path (front end -> DWT -> Tier-1 -> Tier-2 -> codestream
assembled in HBM) over one 8192x8192 frame that is already resident in HBM.  With N GPUs the job is
one image of N tiles of 8192x8192 (image 8192 x 8192*N, tile size 8192): rank r encodes tile r
(weak scaling, fixed work per GPU; at N=1 this is exactly the untiled 8K frame) and the tile-parts are
gathered on rank 0 over RCCL/xGMI -- the path's only exchange step.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# one hardware queue per stream of an encoder handle (main + 4 coder streams); more queues measurably slow the
# chains of short dependent launches (DWT levels) on this ROCm release.  Must be set before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from j2k_amd import api, sharding, synth  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)


class DevView:
    """Zero-copy torch view of a raw device pointer (plumbing for the RCCL gather)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = dict(shape=(nbytes,), typestr="|u1", data=(ptr, False), version=2)


def pmc_traffic(size, prec, levels):
    """HBM bytes of the dominant DWT launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate
    rocprofv3 --pmc runs of this same workload, tools/dwt_pmc.sh -> profiles/r2_dwt_pmc.json).  PMC counters
    cannot be read from inside the timed process, so this is a replayed measurement: the JSON line names its
    source, and other workloads get null."""
    for name in ("r4_dwt_pmc.json", "r3_dwt_pmc.json", "r2_dwt_pmc.json", "r1_dwt_pmc.json"):
        path = os.path.join(ROOT, "profiles", name)
        if (size, prec, levels) == (8192, 16, 5) and os.path.exists(path):
            with open(path) as f:
                d = json.load(f)
            if "fused_hbm_bytes" in d:
                return round(d["fused_hbm_bytes"]), "profiles/" + name
    return None, None


def t1_valu_instructions(size, prec, levels):
    """VALU wave-instructions per frame of the Tier-1 kernels (modeller + coder) from the committed SQ-counter run
    (tools/t1_pmc.sh -> profiles/*_t1_pmc.json); a replayed measurement like pmc_traffic."""
    for name in ("r4_t1_pmc.json", "r3_t1_pmc.json", "r2_t1_pmc.json"):
        path = os.path.join(ROOT, "profiles", name)
        if (size, prec, levels) == (8192, 16, 5) and os.path.exists(path):
            with open(path) as f:
                d = json.load(f)
            if "valu_per_frame" in d:
                return float(d["valu_per_frame"]), "profiles/" + name
    return None


def cpu_baseline(size: int, prec: int, numres: int, seed: int, budget_s: float = 30.0):
    """Reported (not targeted) CPU baseline on this box's host cores: the reference's OpenJPEG
    call sequence (oracle/opj_replay.c over the libopenjp2 found here) on a bounded crop of the same workload,
    timed from opj_setup_encoder to opj_end_compress into a memory sink.  Two figures, each the best of 3
    after a warm-up: 1 thread (what the reference does: its opj_codec_set_threads call is commented out,
    j2k_openjpeg_codec.cpp:624) and opj_codec_set_threads(all cores).  Falls back to the plain-C oracle port."""
    import platform
    from oracle.oracle import Oracle, OpjReplay, make_params
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    model = platform.processor() or ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    side = 4096
    full = synth.planes(size, size, 3, prec, seed)   # the metric frame itself (same generator, same seed)
    pl = np.ascontiguousarray(full[:, :side, :side])  # 1 thread: a bounded crop of it (the full frame would take ~30 s per run)
    p = make_params(side, side, 3, prec, reversible=False, mct=True, numres=numres)
    pf = make_params(size, size, 3, prec, reversible=False, mct=True, numres=numres)
    warm = pl[:, :1024, :1024].copy()
    pw = make_params(1024, 1024, 3, prec, reversible=False, mct=True, numres=numres)
    try:
        rep = OpjReplay()
    except OSError:
        o = Oracle()
        o.encode(warm, pw)
        t0 = time.time()
        o.encode(pl, p)
        secs = time.time() - t0
        return dict(value=round(side * side / secs / 1e6, 3), unit="Mpixels/s", cores=1, kind="port", cpu=model,
                    sample=f"{side}x{side} crop of the same {prec}-bit RGB 9/7 {numres - 1}-level workload, one run of "
                           f"{secs:.1f} s, oracle/j2k_oracle.c (plain-C restatement, 1 thread); no libopenjp2 on this box")

    def best_of(img, par, threads, runs, budget):
        rep.encode(warm, pw, threads=threads)
        times = []
        t_start = time.time()
        for _ in range(runs):
            rep.encode(img, par, threads=threads)
            times.append(rep.last_seconds)
            if time.time() - t_start > budget:  # bounded: a slow box gets fewer runs, and says so
                break
        return min(times), len(times)
    # 1 thread, what the reference does: ONE run of the full frame (the same generated input as the timed workload, SURVEY 8d;
    # ~30 s), and a run of its 4096^2 top-left crop beside it (rounds 1-3 reported the crop)
    sc, _ = best_of(pl, p, 0, 1, budget_s * 0.3)
    rep.encode(full, pf, threads=0)
    s1 = rep.last_seconds
    sn, nn = best_of(full, pf, ncpu, 3, budget_s * 0.6) if ncpu > 1 else (s1, 0)
    what = f"libopenjp2 {rep.version} through the reference's call sequence (opj_setup_encoder..opj_end_compress, memory sink)"
    return dict(value=round(size * size / s1 / 1e6, 3), unit="Mpixels/s", cores=1, kind="reference", cpu=model,
                library=f"libopenjp2 {rep.version}",
                crop=dict(value=round(side * side / sc / 1e6, 3), unit="Mpixels/s", cores=1, sample=f"{side}x{side} top-left crop of the frame, one run: {sc:.2f} s"),
                all_cores=dict(value=round(size * size / sn / 1e6, 3), unit="Mpixels/s", cores=ncpu, runs=nn,
                               sample=f"the full {size}x{size} frame of the timed workload (same generator, same seed)",
                               note="opj_codec_set_threads(cores); the reference leaves it commented out"),
                sample=f"1 thread: the full {size}x{size} {prec}-bit RGB 9/7 {numres - 1}-level frame of the timed workload (same generator, same seed), one run "
                       f"after a 1024^2 warm-up: {s1:.1f} s; all cores: the same frame, best of {nn}: {sn:.2f} s at {ncpu} threads; {what}")


def host_path(api, frame, lay, params, S, frames=6):
    """The plug-in boundary as the host sees it, PCIe included: a pageable host frame goes through the C ABI
    (j2k_hip_encode = what HipCodec::WriteFile calls) and the codestream arrives in a host sink that copies it
    (OutputFile::Write).  Never part of `value`.  Three ways of driving it: one host thread, frame after frame
    (the reference's synchronous WriteFile); four host threads with a handle each (After Effects renders frames
    in parallel); one host thread pipelining three handles with j2k_hip_encode_begin / _end, and with
    j2k_hip_encode_begin_borrowed (the frame stays the library's to read until _end: the upload runs on a thread of the
    handle while the calling thread feeds another handle's sink)."""
    import ctypes as C
    import threading
    planes = api.planes_from_layout(frame.ctypes.data, lay, 3)
    cap = frame.nbytes
    out = {}

    def make_sink(copying=True):
        # the sinks live in the library (j2k_hip_debug_copy_sink: one memcpy per piece into `buf`; _count_sink: a counter):
        # nothing of the interpreter is in the write path
        L = api.load_library()
        fn = api.native_sink(L, copying)
        if copying:
            buf = np.empty(cap, dtype=np.uint8)
            buf[::4096] = 0  # pages touched: a host's output buffer is not fresh memory on every frame
            st = api.CopySink(buf.ctypes.data, cap, 0)
            return fn, C.cast(C.pointer(st), C.c_void_p), (st, buf)
        cnt = C.c_size_t(0)
        return fn, C.cast(C.pointer(cnt), C.c_void_p), (cnt, None)

    def sink_reset(state):
        if isinstance(state[0], api.CopySink):
            state[0].pos = 0
        else:
            state[0].value = 0

    def sink_bytes(state):
        return int(state[0].pos if isinstance(state[0], api.CopySink) else state[0].value)

    sink_cache = {}

    def sync_frames(e, count, copying=True):
        # (one sink per handle and kind, made -- and its pages touched -- before any timing: a host's output buffer is not fresh
        #  memory on every call; rounds 1-3 allocated 512 MB inside the timed region, ~7 ms per frame of page faults)
        key = (id(e), copying)
        if key not in sink_cache:
            sink_cache[key] = make_sink(copying)
        fn, user, state = sink_cache[key]
        for _ in range(count):
            sink_reset(state)
            e._check(e.L.j2k_hip_encode(e.h, C.byref(params), planes, fn, user))
        return sink_bytes(state)

    encs = [api.Encoder(torch.cuda.current_device()) for _ in range(4)]
    try:
        for e in encs:
            sync_frames(e, 1)  # warm-up: arenas, pinned pieces, the sinks
            sync_frames(e, 1, copying=False)
        t0 = time.perf_counter()
        nbytes = sync_frames(encs[0], frames)
        dt = time.perf_counter() - t0
        st = encs[0].stats()
        out["sync_1_thread"] = dict(mpix_s=round(S * S * frames / dt / 1e6, 1), ms_per_frame=round(dt / frames * 1e3, 2),
                                    ms_upload=round(st["ms_upload"], 2), ms_download_wait=round(st["ms_download"], 2),
                                    # the call is band-pipelined: the frame goes up in `bands` row bands while
                                    # the GPU works on the bands that have arrived; ms_after_upload = what the upload did not hide
                                    bands=int(st["bands"]), ms_after_upload=round(st["ms_after_upload"], 2),
                                    early_download_mb=round(st["early_download_bytes"] / 1e6, 1))
        t0 = time.perf_counter()
        sync_frames(encs[0], frames, copying=False)
        dt = time.perf_counter() - t0
        st = encs[0].stats()
        out["sync_1_thread_counting_sink"] = dict(mpix_s=round(S * S * frames / dt / 1e6, 1), ms_per_frame=round(dt / frames * 1e3, 2),
                                                  ms_upload=round(st["ms_upload"], 2), ms_after_upload=round(st["ms_after_upload"], 2), bands=int(st["bands"]),
                                                  note="sink only counts the bytes (no host copy of the 325 MB codestream)")
        # the same call in one piece (upload, then the GPU, then the download: round 3's form), for comparison
        api.tune("bands", -1)
        try:
            sync_frames(encs[0], 1, copying=False)
            t0 = time.perf_counter()
            sync_frames(encs[0], frames, copying=False)
            dt = time.perf_counter() - t0
        finally:
            api.tune("bands", 0)
        out["sync_1_thread_counting_sink_one_piece"] = dict(mpix_s=round(S * S * frames / dt / 1e6, 1), ms_per_frame=round(dt / frames * 1e3, 2))
        for copying, key in ((True, "sync_4_threads"), (False, "sync_4_threads_counting_sink")):
            t0 = time.perf_counter()
            ths = [threading.Thread(target=sync_frames, args=(e, frames, copying)) for e in encs]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            dt = time.perf_counter() - t0
            out[key] = dict(mpix_s=round(S * S * frames * 4 / dt / 1e6, 1), ms_per_frame=round(dt / (frames * 4) * 1e3, 2))
        # one thread, three handles, begin/end
        total = frames * 3
        for copying, key, begin in ((True, "pipelined_1_thread", "j2k_hip_encode_begin"), (False, "pipelined_1_thread_counting_sink", "j2k_hip_encode_begin"),
                                    (True, "pipelined_1_thread_borrowed", "j2k_hip_encode_begin_borrowed"),
                                    (False, "pipelined_1_thread_borrowed_counting_sink", "j2k_hip_encode_begin_borrowed")):
            sinks = [make_sink(copying) for _ in range(3)]
            t0 = time.perf_counter()
            for i in range(total + 2):  # frame i begins on the handle frame i - 3 has left; then frame i - 2 ends: three in flight
                if i < total:
                    k = i % 3
                    encs[k]._check(getattr(encs[k].L, begin)(encs[k].h, C.byref(params), planes))
                if i >= 2:
                    k = (i - 2) % 3
                    sink_reset(sinks[k][2])
                    encs[k]._check(encs[k].L.j2k_hip_encode_end(encs[k].h, sinks[k][0], sinks[k][1]))
            dt = time.perf_counter() - t0
            out[key] = dict(mpix_s=round(S * S * total / dt / 1e6, 1), ms_per_frame=round(dt / total * 1e3, 2),
                            handles=3, api=begin + "/_end")
        out["codestream_bytes"] = int(nbytes)
        out["note"] = ("pageable host frame -> C ABI -> host sink, PCIe both ways included; one handle per thread; the default sink copies the "
                       "codestream in native code (j2k_hip_debug_copy_sink: 325 MB per frame, one memcpy per 32 MiB piece -- what OutputFile::Write into a memory file costs), the "
                       "counting sink only counts; reported beside `value`, never inside it")
    finally:
        for e in encs:
            e.close()
    return out


def decode_path(api, cs: bytes, planes, S, device):
    """SURVEY 8f N4 beside the headline: the timed configuration's own codestream decoded again (host file bytes -> planar host
    channels kept from call to call, PCIe both ways included), one frame at a time and with three host threads; for the
    reversible transform the samples must be the frame's, for 9/7 they must equal the device decode into a device buffer
    (the oracle's decode of an 8K frame does not fit a bench run: the golden tests hold the decoder to libopenjp2)."""
    import threading
    out = {}
    e = api.Encoder(device)
    try:
        buf = None
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            buf = e.decode_planar(cs, out=buf)
            ts.append(time.perf_counter() - t0)
        st = e.stats()
        out["one_frame_ms"] = round(min(ts[1:]) * 1e3, 2)
        out["mpix_s"] = round(S * S / min(ts[1:]) / 1e6, 1)
        out["stages_ms"] = {k: round(st[k], 2) for k in ("ms_t2_host", "ms_upload", "ms_t1", "ms_dwt", "ms_frontend")}
        out["stages_note"] = "ms_t1 = gather + Tier-1 decode, ms_dwt = inverse DWT, ms_frontend = inverse MCT + output stage"
        ref = buf.copy()
    finally:
        e.close()
    nt, per = 3, 3
    encs = [api.Encoder(device) for _ in range(nt)]
    outs = [None] * nt
    try:
        def work(i, n):
            for _ in range(n):
                outs[i] = encs[i].decode_planar(cs, out=outs[i])
        ths = [threading.Thread(target=work, args=(i, 1)) for i in range(nt)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        t0 = time.perf_counter()
        ths = [threading.Thread(target=work, args=(i, per)) for i in range(nt)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        dt = time.perf_counter() - t0
        out["in_flight"] = dict(threads=nt, ms_per_frame=round(dt / (nt * per) * 1e3, 2), mpix_s=round(S * S * nt * per / dt / 1e6, 1))
        out["consistent"] = bool(all(np.array_equal(o, ref) for o in outs))
    finally:
        for x in encs:
            x.close()
    psnr = None
    if planes is not None:  # distance of the decoded frame from the source (9/7 at full rate: the quantisation's)
        d = ref.astype(np.float64) - planes.astype(np.float64)
        mse = float(np.mean(d * d))
        peak = float((1 << (16 if ref.dtype == np.uint16 else 8)) - 1)
        psnr = None if mse == 0 else round(10.0 * np.log10(peak * peak / mse), 2)
    out["psnr_db_vs_source"] = psnr
    out["note"] = "decode of the timed configuration's codestream (j2k_hip_decode: what HipCodec::ReadFile calls); reported beside `value`, never inside it"
    return out


def rate_control_path(api, planes, S, prec, numres, device, ratio=20.0, inflight=5, per=6):
    """SURVEY 8f N2 beside the headline: the same resident frame encoded to a byte budget (one layer, compression ratio
    `ratio`, OpenJPEG's cp_disto_alloc semantics).  Tier-1 also produces per-pass byte counts and distortion sums; the
    passes are allocated by OpenJPEG's bisection on the host with the per-block work on the device (rate_control.cpp,
    rate.hip) -- reported next to `value`, never inside it."""
    import ctypes as C
    import threading
    p = api.make_params(S, S, 3, prec, reversible=False, ycc=True, num_resolutions=numres, comment="", rates=[ratio])
    encs = [api.Encoder(device) for _ in range(inflight)]
    outs = [(C.c_void_p(), C.c_size_t()) for _ in range(inflight)]

    def one(k):
        e = encs[k]
        e._check(e.L.j2k_hip_encode_device(e.h, C.byref(p), planes, C.byref(outs[k][0]), C.byref(outs[k][1]), None, 0))

    for k in range(inflight):
        one(k)  # warm: arenas, plans
    one(0)
    n, t1s, hosts = 4, [], []
    t0 = time.perf_counter()
    for _ in range(n):
        one(0)
        st = encs[0].stats()
        t1s.append(st["ms_t1"]); hosts.append(st["ms_t2_host"])
    alone = (time.perf_counter() - t0) / n
    ths = [threading.Thread(target=lambda k=k: [one(k) for _ in range(per)]) for k in range(inflight)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = (time.perf_counter() - t0) / (per * inflight)
    sizes = {int(o[1].value) for o in outs}
    for e in encs:
        e.close()
    return {"layer_ratio": ratio, "codestream_bytes": sizes.pop() if len(sizes) == 1 else sorted(sizes),
            "achieved_ratio": round(S * S * 3 * ((prec + 7) // 8) / max(1, int(outs[0][1].value)), 3),
            "one_frame_ms": round(alone * 1e3, 2), "ms_t1": round(sum(t1s) / n, 2), "ms_host_allocation_and_tier2": round(sum(hosts) / n, 2),
            "frames_in_flight": inflight, "ms_per_frame": round(dt * 1e3, 3), "mpix_s": round(S * S / dt / 1e6, 1),
            "allocation": "per-block work (distortions, bounds, scans of rounds with >= 512 open blocks, candidate sums) on the device (rate.hip), "
                          "OpenJPEG's bisection and the exact pricing of the last candidates on the host (DESIGN.md 11)",
            "note": "byte-identical to libopenjp2 under the same tcp_rates (tests/test_rate_control.py); not part of `value`"}


def run_config_mode(args, rank, local_rank, world, backend):
    """The two multi-GPU configurations of BASELINE.json beside the metric's weak-scaling run (both strong scaling:
    the job is fixed, the ranks share it):
      c4  one 16384x16384 16-bit RGB image, tiles of 2048^2 (64 tiles), 5/3 + RCT: rank r encodes a contiguous block of
          tile rows (8 tiles per rank at N = 8) and the tile-parts are gathered on rank 0 over RCCL;
      c5  64 independent frames of 4096x2160 10-bit RGB, 9/7 + ICT (an image sequence): frame f goes to rank f % N, the
          frames of a rank go through j2k_hip_encode_sequence_device 8 at a time; no exchange at all (one file per frame).
    A step = the whole job once.  The rank's input is resident in HBM when the timed region starts."""
    import ctypes as C
    import hashlib
    import threading
    mode = args.mode
    if mode == "c4":
        W = H = 16384
        T, prec = 2048, 16
        ntiles_x = W // T
        rows_t = sharding.partition_tiles(H // T, world)[rank]  # tile rows of this rank
        nrows = rows_t[1] * T
        params = api.make_params(W, H, 3, prec, reversible=True, ycc=True, tile_size=T, comment="")
        frame = lay = None
        d_frame = None
        if nrows:
            pl = synth.planes(W, nrows, 3, prec, 34567 + rank)
            frame, lay = synth.ae_frame(pl, prec)
            del pl
            d_frame = torch.from_numpy(frame).cuda()
            del frame
        else:
            lay = synth.ae_frame(synth.planes(8, 8, 3, prec, 1), prec)[1]
            lay["rowbytes"] = 8 * W
        base = (d_frame.data_ptr() if nrows else 0) - rows_t[0] * T * lay["rowbytes"]
        planes = api.planes_from_layout(base, lay, 3)
        nfl = max(1, args.inflight)
        encs = [api.Encoder(local_rank) for _ in range(nfl)]
        outs = [(C.c_void_p(), C.c_size_t()) for _ in range(nfl)]
        recv = [None]

        def one(slot):
            e = encs[slot]
            if rows_t[1]:
                e._check(e.L.j2k_hip_encode_tiles_device(e.h, C.byref(params), planes, rows_t[0] * ntiles_x, rows_t[1] * ntiles_x,
                                                         C.byref(outs[slot][0]), C.byref(outs[slot][1]), None, 0))
        lock = threading.Lock()

        def exchange(slot):
            n = outs[slot][1].value if rows_t[1] else 0
            view = torch.as_tensor(DevView(outs[slot][0].value, n), device="cuda") if n else torch.empty(0, dtype=torch.uint8, device="cuda")
            if backend != "nccl":
                view = view.cpu()
            with lock:  # the collective calls of a rank must come in frame order
                _, recv[0] = sharding.gather_tileparts(view, rank, world, recv[0])

        def run(count):
            # frames in flight: handle k encodes step i = k, k + nfl, ...; the gather of step i runs in step order
            order = threading.Condition()
            turn = [0]

            def worker(k):
                torch.cuda.set_device(local_rank)
                for i in range(k, count, nfl):
                    one(k)
                    if world > 1:
                        with order:
                            order.wait_for(lambda: turn[0] == i)
                        exchange(k)
                        with order:
                            turn[0] += 1
                            order.notify_all()
            ths = [threading.Thread(target=worker, args=(k,)) for k in range(nfl)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        pixels = W * H
        workload = (f"one {W}x{H} {prec}-bit RGB image, tiles {T}x{T} (64 tiles), 5/3 reversible + RCT; {64 // max(world, 1)} tiles per rank "
                    f"(contiguous tile rows), tile-parts gathered on rank 0 over RCCL; AE ARGB64 rows resident in HBM")
        cs_bytes = lambda: int(outs[0][1].value)
    else:
        W, H, prec, NF, PER = 4096, 2160, 10, 64, 8
        mine = list(range(rank, NF, world))
        params = api.make_params(W, H, 3, prec, reversible=False, ycc=True, comment="")
        boot = api.Encoder(local_rank)
        lay = None
        dptrs = []
        keep = []
        for f in mine:
            fr, lay = synth.ae_frame(synth.planes(W, H, 3, prec, 45678 + f), prec)
            t = torch.from_numpy(fr).cuda()
            keep.append(t)
            dptrs.append(t.data_ptr())
        nfl = max(1, min(args.inflight, max(1, len(mine) // PER)))
        encs = [api.Encoder(local_rank) for _ in range(nfl)]
        calls = [mine[i:i + PER] for i in range(0, len(mine), PER)]
        last = [None] * nfl

        def run(count):
            def worker(k):
                torch.cuda.set_device(local_rank)
                for j in range(k, count * len(calls), nfl):
                    idx = j % len(calls)
                    lo = idx * PER
                    last[k] = encs[k].encode_sequence_device(dptrs[lo:lo + len(calls[idx])], lay, params, download=False)
            ths = [threading.Thread(target=worker, args=(k,)) for k in range(nfl)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        pixels = W * H * NF
        workload = (f"{NF} frames {W}x{H} {prec}-bit RGB (16-bit container), 9/7 irreversible + ICT, 6 resolutions; frame f on rank f % N, "
                    f"{PER} frames per j2k_hip_encode_sequence_device call, {nfl} calls in flight per GPU; frames resident in HBM, one codestream per frame, no exchange")
        cs_bytes = lambda: int(sum(x[1] for x in last[0])) if last[0] else 0

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    steps, warm = max(1, args.steps), max(1, args.warmup)
    run(max(warm, nfl))  # (every handle in flight takes at least one untimed step: its arenas are allocated on first use)
    fence()
    t0 = time.perf_counter()
    run(steps)
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # ---- after the timed region: every handle's last output of the timed configuration is hashed and compared with a
    # re-encode of the same input on an idle chip (one handle, nothing else in flight) and, where tests/golden holds
    # libopenjp2's hash for exactly this input (c4 at N = 1: the whole 64-tile image; c5: frame 0), with that.
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        gold = json.load(f)
    sha = lambda b: hashlib.sha256(b).hexdigest()
    verified, against = True, []
    if mode == "c4":
        timed = [sha(encs[k].d2h(outs[k][0].value, outs[k][1].value)) if rows_t[1] and outs[k][1].value else None for k in range(min(nfl, steps))]
        if rows_t[1]:
            one(0)
            idle = encs[0].d2h(outs[0][0].value, outs[0][1].value)
            verified = all(h == sha(idle) for h in timed)
            against.append("idle-chip re-encode of the rank's tile rows")
            g = gold.get("c4_16384_rgb16_53_tile2048")
            if world == 1 and g:
                whole = api.main_header(params) + idle.tobytes() + b"\xff\xd9"
                verified = verified and len(whole) == g["length"] and sha(whole) == g["sha256"]
                against.append("tests/golden/golden.json c4_16384_rgb16_53_tile2048 (libopenjp2, sha256 of the whole codestream)")
    else:
        for k in range(min(nfl, max(1, steps * len(calls)))):
            if last[k] is None:
                continue
            # which call produced last[k]: the worker of slot k ran j = k, k + nfl, ... < steps * len(calls)
            total = steps * len(calls)
            j_last = k + ((total - 1 - k) // nfl) * nfl
            lo = (j_last % len(calls)) * PER
            for i, (dptr, n, _) in enumerate(last[k]):
                h_timed = sha(encs[k].d2h(dptr, n))
                d1, n1, _ = boot.encode_device(dptrs[lo + i], lay, params, download=False)
                h_idle = sha(boot.d2h(d1, n1))
                verified = verified and h_timed == h_idle
                if mine[lo + i] == 0 and "c5_frame0_4096x2160_rgb10_97" in gold:
                    g = gold["c5_frame0_4096x2160_rgb10_97"]
                    verified = verified and n1 == g["length"] and h_idle == g["sha256"]
                    against.append("tests/golden/golden.json c5_frame0_4096x2160_rgb10_97 (libopenjp2, sha256) for frame 0")
        against.insert(0, "idle-chip single-frame re-encode of every frame of each handle's last sequence call")
        boot.close()
    v = torch.tensor([1 if verified else 0], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
    if not verified:
        print(f"rank {rank}: a codestream of the timed configuration differs from its idle-chip re-encode / golden", file=sys.stderr, flush=True)
    verified = bool(v.item())
    if rank == 0:
        print(json.dumps({
            "verified": verified, "verified_against": "; ".join(against),
            "metric": f"Mpixels/s encode, BASELINE config {mode.upper()}", "value": round(pixels * steps / elapsed / 1e6, 2), "unit": "Mpixels/s",
            "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "int32" if mode == "c4" else "f32", "data": "synthetic",
            "config": {"workload": workload, "codestream_bytes_rank0": cs_bytes(), "parallelism": ("tile-sharded" if mode == "c4" else "frame-sharded") + f" x{world}"},
        }), flush=True)
    for e in encs:
        e.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--size", type=int, default=8192, help="frame side (default: the metric's 8192)")
    ap.add_argument("--prec", type=int, default=16)
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive plug-in-boundary measurement")
    ap.add_argument("--no-decode", action="store_true", help="skip the decode of the timed codestream (reported beside value)")
    ap.add_argument("--no-rate-control", action="store_true", help="skip the rate-controlled encode of the same frame (reported beside value)")
    ap.add_argument("--no-dwt-replay", action="store_true",
                    help="skip the replays of the DWT launches after the timed region (roofline.phase.sum_kernel / alone_back_to_back): "
                         "tools/profile_round.sh passes it so that rocprofv3 --stats averages the launches of the timed region only")
    ap.add_argument("--mode", choices=["c3", "c4", "c5"], default="c3",
                    help="c3 (default): the metric's frame, weak scaling; c4 / c5: the multi-GPU configurations of BASELINE.json, strong scaling")
    ap.add_argument("--inflight", type=int, default=3,
                    help="frames in flight per GPU: independent encoder handles driven by host threads, so that one frame's "
                         "MQ-coder tail and host Tier-2 overlap the next frame's DWT/modelling (image-sequence path)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # rehearsal knobs (1-GPU boxes): J2K_BENCH_SHARE_GPU=1 puts every rank on device 0,
    # J2K_BENCH_BACKEND=gloo gathers through host memory instead of RCCL
    if os.environ.get("J2K_BENCH_SHARE_GPU"):
        local_rank = 0
    backend = os.environ.get("J2K_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    if args.mode != "c3":
        if args.steps == 240 and args.warmup == 6:
            args.steps, args.warmup = 8, 2
        return run_config_mode(args, rank, local_rank, world, backend)

    S, prec, numres = args.size, args.prec, args.levels + 1
    W, H = S, S * world  # one S x S tile per rank
    seed = 23456 + rank
    pl = synth.planes(S, S, 3, prec, seed)
    frame, lay = synth.ae_frame(pl, prec)
    del pl
    d_frame = torch.from_numpy(frame).cuda()  # the rank's tile rows, resident in HBM
    del frame
    # channel views describe the whole image; only the rows of this rank's tile exist (and are read)
    base = d_frame.data_ptr() - rank * S * lay["rowbytes"]
    params = api.make_params(W, H, 3, prec, reversible=False, ycc=True, num_resolutions=numres,
                             tile_size=S if world > 1 else 0, comment="")
    import ctypes as C
    import threading
    nfl = max(1, args.inflight)
    encs = [api.Encoder(local_rank) for _ in range(nfl)]
    enc = encs[0]
    planes = api.planes_from_layout(base, lay, 3)
    outs = [(C.c_void_p(), C.c_size_t()) for _ in range(nfl)]

    def stage_tilepart(payload, slot_buf):
        """A frame's tile-part out of the encoder's buffer (which the handle's next frame overwrites) into a staging buffer."""
        dptr, n = payload
        view = torch.as_tensor(DevView(dptr, n), device="cuda")
        if backend != "nccl":
            return view.cpu(), slot_buf
        if slot_buf is None or slot_buf.numel() < n:
            slot_buf = torch.empty(int(n * 1.1) + 4096, dtype=torch.uint8, device="cuda")
        staged = slot_buf[:n]
        staged.copy_(view)
        torch.cuda.current_stream().synchronize()
        return staged, slot_buf

    exchange = sharding.Exchange(rank, world, nfl + 1, stage_tilepart, setup=lambda: torch.cuda.set_device(local_rank)) if world > 1 else None

    def step(slot=0, frame=0):
        e = encs[slot]
        dptr, n = outs[slot]
        if world == 1:
            e._check(e.L.j2k_hip_encode_device(e.h, C.byref(params), planes, C.byref(dptr), C.byref(n), None, 0))
            return
        e._check(e.L.j2k_hip_encode_tiles_device(e.h, C.byref(params), planes, rank, 1, C.byref(dptr), C.byref(n), None, 0))
        exchange.submit(frame, (dptr.value, n.value))

    def run_steps(count):
        """`count` frames through `nfl` encoder handles (frame i on handle i % nfl); returns per-frame stats of slot 0."""
        stats = []
        if exchange:
            exchange.reset()
        if nfl == 1:
            for i in range(count):
                step(0, i)
                stats.append((encs[0].stats(), encs[0].dwt_level_ms()))
            if exchange:
                exchange.drain(count)
            return stats
        def worker(slot):
            torch.cuda.set_device(local_rank)
            for i in range(slot, count, nfl):
                step(slot, i)
                if slot == 0:
                    stats.append((encs[0].stats(), encs[0].dwt_level_ms()))
        ths = [threading.Thread(target=worker, args=(k,)) for k in range(nfl)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if exchange:
            exchange.drain(count)  # the step is done when rank 0 holds every tile-part
        return stats

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(max(args.warmup, nfl if args.warmup else 0))
    fence()
    dwt_ms, dwt_bytes, stage = [], 0.0, {}
    t0 = time.perf_counter()
    per_frame = run_steps(args.steps)
    fence()
    for st, lv in per_frame:
        dwt_ms.append(lv)
        dwt_bytes = st["dwt_bytes"]
        for k in ("ms_frontend", "ms_dwt", "ms_t1", "ms_t2_host", "ms_assemble", "ms_total"):
            stage[k] = stage.get(k, 0.0) + st[k] / len(per_frame)
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # ---- after the timed region (none of this is inside `value`) ------------------------------------
    # (1) every handle's last codestream of the timed region -- produced by the timed configuration itself:
    # frames in flight, two coder groups, held-back and yielding coder launches -- is hashed and compared with
    # libopenjp2's codestream for this workload (tests/golden/golden.json; same seed, no COM segment).  Other
    # workloads (other sizes, tile-sharded ranks) are compared with a re-encode on an idle chip instead.
    import hashlib
    timed_hashes = [hashlib.sha256(encs[k].d2h(outs[k][0].value, outs[k][1].value)).hexdigest()
                    for k in range(min(nfl, args.steps))]
    verified, verified_against = None, None
    if world == 1 and (S, prec, args.levels) == (8192, 16, 5):
        with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
            gold = json.load(f)["c3_8192_rgb16_97_5lvl"]
        verified = all(h == gold["sha256"] for h in timed_hashes) and int(outs[0][1].value) == gold["length"]
        verified_against = "tests/golden/golden.json c3_8192_rgb16_97_5lvl (libopenjp2 %s, sha256)" % "2.4.0/2.5.4"
    # (2) the same DWT launches with nothing else on the chip (one frame at a time, one handle), reported
    # beside the live figure as roofline.alone
    alone_lv = []
    if exchange:
        exchange.reset()
    for k in range(3):
        step(0, k)
        if exchange:
            exchange.drain(k + 1)
        alone_lv.append(encs[0].dwt_level_ms())
    fence()
    # the whole DWT phase alone: the frame's launches replayed back to back between two events (an event between
    # two dependent launches costs ~20 us of queue time, which the 7-25 us launches of levels 3-5 would carry)
    alone_replay, level_kernel_ms = None, None
    if not args.no_dwt_replay and alone_lv and len(alone_lv[0]) > 0:
        alone_replay = (encs[0].dwt_time(0, 1, 20), encs[0].dwt_time(0, len(alone_lv[0]), 20))
        # SURVEY 8d's phase figure is Sigma bytes / Sigma KERNEL time: every level's launch replayed 20 times back to back on its own
        # (no event packet between dependent launches, no queue latency inside the bracket)
        level_kernel_ms = [encs[0].dwt_time(l, 1, 20) for l in range(len(alone_lv[0]))]
    alone_cs = encs[0].d2h(outs[0][0].value, outs[0][1].value).tobytes()  # (kept: the decode leg reads this very codestream back)
    alone_hash = hashlib.sha256(alone_cs).hexdigest()
    if verified is None:
        verified = all(h == alone_hash for h in timed_hashes)
        verified_against = "re-encode of the same input on an idle chip (no golden for this workload)"
    if world > 1:
        v = torch.tensor([1 if verified else 0], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        verified = bool(v.item())
    if not verified:
        print(f"rank {rank}: codestream of the timed configuration differs: {timed_hashes} vs {alone_hash}", file=sys.stderr, flush=True)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = (S * S * world * args.steps) / elapsed / 1e6
        # Roofline of the dominant kernel = the level-1 launch (front end fused in: three quarters of the DWT's
        # algorithmic bytes).  Algorithmic bytes of one launch (one read + one write of the level's region at
        # 4 B/sample) / mean launch duration (hipEvents on the encoder's stream right before and
        # after the launch, averaged over the timed steps).  `phase` = all levels together, the way SURVEY 8d
        # defines the DWT figure (Sigma bytes / Sigma launch time).
        nl = len(dwt_ms[0]) if dwt_ms else 0
        l1_bytes = 8.0 * 3 * S * S
        l1_ms = float(np.mean([x[0] for x in dwt_ms])) if nl else 0.0
        phase_ms = float(np.mean([sum(x) for x in dwt_ms])) if nl else 0.0
        gbps = lambda nbytes, ms: nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic(S, prec, args.levels)
        a1 = float(np.median([x[0] for x in alone_lv])) if nl else 0.0
        ap = float(np.median([sum(x) for x in alone_lv])) if nl else 0.0
        out = {
            "metric": "Mpixels/s encode, 8Kx8K 16-bit RGB, 5 DWT levels; DWT HBM GB/s vs roofline",
            "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "verified": bool(verified), "verified_against": verified_against,
            "config": {"workload": f"{S}x{S} {prec}-bit RGB per GPU, 9/7 irreversible + ICT, {args.levels} DWT levels, "
                                   f"64x64 code-blocks, AE ARGB64 frame resident in HBM -> codestream assembled in HBM"
                                   + ("" if world == 1 else f"; image {W}x{H}, one {S}x{S} tile per rank, tile-parts gathered on rank 0 over RCCL"),
                       "distribution": "A (gradient + (prec-4)-bit LCG noise, SURVEY 8d)", "seed": 23456,
                       "codestream_bytes": int(outs[0][1].value), "parallelism": f"tile-sharded x{world}",
                       "frames_in_flight": nfl},
            "roofline": {"bound": "hbm", "kernel": "dwt_fused_kernel<false,3> (9/7 level 1 with the sample front end fused in)",
                         "achieved": round(gbps(l1_bytes, l1_ms), 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(gbps(l1_bytes, l1_ms) / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "bytes_per_launch": l1_bytes, "mean_launch_ms": round(l1_ms, 4),
                         "alone": {"achieved": round(gbps(l1_bytes, a1), 1), "frac": round(gbps(l1_bytes, a1) / HBM_PEAK_GBPS, 4),
                                   "mean_launch_ms": round(a1, 4)},
                         "phase": {"launches": nl, "bytes": dwt_bytes, "ms": round(phase_ms, 4),
                                   "achieved": round(gbps(dwt_bytes, phase_ms), 1), "frac": round(gbps(dwt_bytes, phase_ms) / HBM_PEAK_GBPS, 4),
                                   "alone": {"ms": round(ap, 4), "achieved": round(gbps(dwt_bytes, ap), 1),
                                             "frac": round(gbps(dwt_bytes, ap) / HBM_PEAK_GBPS, 4)},
                                   "sum_kernel": None if not level_kernel_ms else {
                                       "per_launch_ms": [round(x, 4) for x in level_kernel_ms], "ms": round(sum(level_kernel_ms), 4),
                                       "achieved": round(gbps(dwt_bytes, sum(level_kernel_ms)), 1),
                                       "frac": round(gbps(dwt_bytes, sum(level_kernel_ms)) / HBM_PEAK_GBPS, 4),
                                       "note": "Sigma bytes / Sigma kernel time (SURVEY 8d): each level's launch replayed back to back on an idle chip "
                                               "(j2k_hip_debug_dwt_time), so no queue latency of event packets is inside"},
                                   "alone_back_to_back": None if not alone_replay else {
                                       "level1_ms": round(alone_replay[0], 4), "ms": round(alone_replay[1], 4),
                                       "achieved": round(gbps(dwt_bytes, alone_replay[1]), 1),
                                       "frac": round(gbps(dwt_bytes, alone_replay[1]) / HBM_PEAK_GBPS, 4)}},
                         "note": ("achieved/frac are live values from the timed region with %d frames in flight (the DWT of one "
                                  "frame runs beside the MQ coder waves of the others); 'alone' = the same launches on an idle "
                                  "chip after the timed region; 'phase' = all %d DWT launches of a frame together.  The dominant "
                                  "launch is timed by HIP events attached to its own dispatch (hipExtLaunchKernelGGL start / stop: the "
                                  "kernel's begin and end on its stream, what rocprofv3 --kernel-trace reports), the phase by event "
                                  "records around its launches" % (nfl, nl))},
            "stages_ms": {k: round(v, 3) for k, v in stage.items()},  # per frame, as seen by one handle (ms_total = frame latency)
        }
        # Tier-1 work (no roofline fraction is claimed for the serial, integer Tier-1 -- its rate of work is reported)
        st0 = per_frame[-1][0] if per_frame else None
        if st0:
            per_s = world * args.steps / elapsed  # frames per second, whole job
            valu = t1_valu_instructions(S, prec, args.levels)
            out["t1"] = {"codeblocks_per_s": round(st0["num_codeblocks"] * per_s, 1),
                         "decisions_per_s": round(st0["num_symbols"] * per_s, 1),
                         "coded_bits_per_s": round(8.0 * int(outs[0][1].value) * per_s, 1),
                         "codeblocks_per_frame": int(st0["num_codeblocks"]), "decisions_per_frame": int(st0["num_symbols"]),
                         "valu_frac": None if not valu else round(valu[0] * 4.0 / (1024 * 2.4e9) * per_s / world, 4),
                         "valu_source": None if not valu else valu[1],
                         "note": "coded bits = bits of the finished codestream; valu_frac = VALU wave-instructions of the modeller + coder per frame "
                                 "(SQ_INSTS_VALU, replayed from the committed PMC run) x 4 cycles / (1024 SIMDs x 2.4 GHz) x frames per second per GPU"}
        if world == 1 and not args.no_host_path:
            for e in encs[1:]:
                e.close()
            hp = synth.planes(S, S, 3, prec, 23456)
            hframe, hlay = synth.ae_frame(hp, prec)
            del hp
            out["host_path"] = host_path(api, hframe, hlay, params, S)
            del hframe
        if world == 1 and not args.no_rate_control:  # (after the host path: its handles and arenas are gone again by then)
            for e in encs:  # the timed region's handles too: their streams would share the hardware queues with the new ones
                e.close()
            out["rate_control"] = rate_control_path(api, planes, S, prec, numres, local_rank)
        if world == 1 and not args.no_decode:
            for e in encs:  # (already closed above unless the other legs were skipped)
                e.close()
            out["decode"] = decode_path(api, alone_cs, synth.planes(S, S, 3, prec, seed), S, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(S, prec, numres, 23456)
        print(json.dumps(out), flush=True)
    if exchange:
        exchange.close()
    for e in encs:
        e.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

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
    """HBM bytes per DWT launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate
    rocprofv3 --pmc runs of this same workload; profiles/r1_dwt_pmc.json), or None for other workloads."""
    path = os.path.join(ROOT, "profiles", "r1_dwt_pmc.json")
    if (size, prec, levels) != (8192, 16, 5) or not os.path.exists(path):
        return None
    with open(path) as f:
        return round(json.load(f)["hbm_bytes_per_launch"])


def cpu_baseline(width: int, prec: int, numres: int, seed: int):
    """Reported (not targeted) CPU baseline on this box's host cores: the reference's OpenJPEG call
    sequence (oracle/opj_replay.c) on a bounded crop of the same workload, single-threaded like the
    reference (j2k_openjpeg_codec.cpp:624).  Falls back to the plain-C oracle port."""
    from oracle.oracle import Oracle, OpjReplay, make_params
    side = 4096
    pl = synth.planes(side, side, 3, prec, seed)
    p = make_params(side, side, 3, prec, reversible=False, mct=True, numres=numres)
    try:
        rep = OpjReplay()
        rep.encode(pl[:, :512, :512].copy(), make_params(512, 512, 3, prec, reversible=False, mct=True, numres=numres))
        rep.encode(pl, p)
        secs = rep.last_seconds
        kind, what = "reference", f"libopenjp2 {rep.version} via the reference's call sequence"
    except OSError:
        o = Oracle()
        t0 = time.time()
        o.encode(pl, p)
        secs = time.time() - t0
        kind, what = "port", "oracle/j2k_oracle.c (plain-C restatement)"
    return dict(value=round(side * side / secs / 1e6, 3), unit="Mpixels/s", cores=1, kind=kind,
                sample=f"{side}x{side} crop of the same {prec}-bit RGB 9/7 {numres - 1}-level workload, "
                       f"{secs:.1f} s, 1 thread, {what}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--size", type=int, default=8192, help="frame side (default: the metric's 8192)")
    ap.add_argument("--prec", type=int, default=16)
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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

    class Exchange:
        """The one exchange step of the path, off the encode threads: a frame's tile-part is copied out of the
        encoder's buffer (which the handle's next frame overwrites) into a staging buffer, and one thread per
        rank gathers the frames on rank 0 strictly in frame order -- the same order on every rank -- while the
        encode threads go on with the next frames."""

        def __init__(self, depth):
            self.cv = threading.Condition()
            self.ready = {}            # frame -> staged tensor
            self.next = 0              # next frame to exchange
            self.free = [None] * depth  # staging buffers (allocated on first use)
            self.avail = list(range(depth))
            self.recv = None
            self.error = None
            self.stop = False
            self.thread = threading.Thread(target=self._run, daemon=True)
            self.thread.start()

        def submit(self, frame, dptr, n):
            with self.cv:
                self.cv.wait_for(lambda: self.avail or self.error)
                if self.error:
                    raise self.error
                slot = self.avail.pop()
            view = torch.as_tensor(DevView(dptr, n), device="cuda")
            if backend != "nccl":
                staged = view.cpu()
            else:
                if self.free[slot] is None or self.free[slot].numel() < n:
                    self.free[slot] = torch.empty(int(n * 1.1) + 4096, dtype=torch.uint8, device="cuda")
                staged = self.free[slot][:n]
                staged.copy_(view)
                torch.cuda.current_stream().synchronize()
            with self.cv:
                self.ready[frame] = (slot, staged)
                self.cv.notify_all()

        def _run(self):
            try:
                torch.cuda.set_device(local_rank)
                while True:
                    with self.cv:
                        self.cv.wait_for(lambda: self.stop or self.next in self.ready)
                        if self.stop and self.next not in self.ready:
                            return
                        slot, staged = self.ready.pop(self.next)
                    _, self.recv = sharding.gather_tileparts(staged, rank, world, self.recv)
                    with self.cv:
                        self.next += 1
                        self.avail.append(slot)
                        self.cv.notify_all()
            except BaseException as ex:  # surfaces in submit()/drain()
                with self.cv:
                    self.error = ex
                    self.cv.notify_all()

        def reset(self):
            with self.cv:
                self.next = 0

        def drain(self, count):
            with self.cv:
                self.cv.wait_for(lambda: self.next >= count or self.error)
                if self.error:
                    raise self.error

        def close(self):
            with self.cv:
                self.stop = True
                self.cv.notify_all()
            self.thread.join()

    exchange = Exchange(nfl + 1) if world > 1 else None

    def step(slot=0, frame=0):
        e = encs[slot]
        dptr, n = outs[slot]
        if world == 1:
            e._check(e.L.j2k_hip_encode_device(e.h, C.byref(params), planes, C.byref(dptr), C.byref(n), None, 0))
            return
        e._check(e.L.j2k_hip_encode_tiles_device(e.h, C.byref(params), planes, rank, 1, C.byref(dptr), C.byref(n), None, 0))
        exchange.submit(frame, dptr.value, n.value)

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
    # after the timed region: the same DWT launches with nothing else on the chip (one frame at a time,
    # one handle), reported beside the live figure as roofline.alone
    alone_ms = []
    if nfl > 1:
        if exchange:
            exchange.reset()
        for k in range(3):
            step(0, k)
            if exchange:
                exchange.drain(k + 1)
            alone_ms.append(sum(encs[0].dwt_level_ms()))
        fence()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = (S * S * world * args.steps) / elapsed / 1e6
        # roofline of the 9/7 DWT kernel: algorithmic bytes of one launch (one read + one write of the
        # level's region at 4 B/sample) / mean launch duration (hipEvents on the
        # encoder's stream around every level launch, averaged over the timed steps and levels)
        nl = len(dwt_ms[0]) if dwt_ms else 0
        mean_launch_ms = float(np.mean([sum(x) for x in dwt_ms]) / max(nl, 1)) if nl else 0.0
        bytes_per_launch = dwt_bytes / max(nl, 1)
        achieved = bytes_per_launch / (mean_launch_ms * 1e-3) / 1e9 if mean_launch_ms > 0 else 0.0
        out = {
            "metric": "Mpixels/s encode, 8Kx8K 16-bit RGB, 5 DWT levels; DWT HBM GB/s vs roofline",
            "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{S}x{S} {prec}-bit RGB per GPU, 9/7 irreversible + ICT, {args.levels} DWT levels, "
                                   f"64x64 code-blocks, AE ARGB64 frame resident in HBM -> codestream assembled in HBM"
                                   + ("" if world == 1 else f"; image {W}x{H}, one {S}x{S} tile per rank, tile-parts gathered on rank 0 over RCCL"),
                       "distribution": "A (gradient + (prec-4)-bit LCG noise, SURVEY 8d)", "seed": 23456,
                       "codestream_bytes": int(outs[0][1].value), "parallelism": f"tile-sharded x{world}",
                       "frames_in_flight": nfl},
            "roofline": {"bound": "hbm", "kernel": "dwt_level_kernel<false> (9/7, one launch per level)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": pmc_traffic(S, prec, args.levels),
                         "bytes_per_launch": bytes_per_launch, "mean_launch_ms": round(mean_launch_ms, 4),
                         "launches_per_step": nl},
            "stages_ms": {k: round(v, 3) for k, v in stage.items()},  # per frame, as seen by one handle (ms_total = frame latency)
        }
        if alone_ms and nl:
            # live = measured over the timed region, where the DWT of one frame runs beside the MQ coder
            # chains of the frames before it; alone = the same launches with the chip to themselves
            a_ms = float(np.median(alone_ms)) / nl
            a_gbps = bytes_per_launch / (a_ms * 1e-3) / 1e9
            out["roofline"]["alone"] = {"achieved": round(a_gbps, 1), "frac": round(a_gbps / HBM_PEAK_GBPS, 4),
                                        "mean_launch_ms": round(a_ms, 4)}
            out["roofline"]["note"] = ("achieved/frac are live values from the timed region with %d frames in flight "
                                       "(DWT co-runs with other frames' MQ coder waves); 'alone' = same launches, idle chip" % nfl)
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

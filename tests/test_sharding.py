"""CPU: the multi-rank path (tile partition + variable-length gather on rank 0 + assembly) over gloo
with world_size 2 and 3.  The tile-parts come from the committed golden codestream (split at its SOT
markers), so the test checks that sharded output reassembles to the reference bytes."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN_DIR, ROOT
from j2k_amd import sharding


def test_partition_tiles():
    assert sharding.partition_tiles(64, 8) == [(i * 8, 8) for i in range(8)]
    assert sharding.partition_tiles(6, 4) == [(0, 2), (2, 2), (4, 1), (5, 1)]
    assert sharding.partition_tiles(2, 3) == [(0, 1), (1, 1), (2, 0)]
    for nt, w in [(1, 1), (7, 2), (16, 8), (5, 8)]:
        parts = sharding.partition_tiles(nt, w)
        assert sum(n for _, n in parts) == nt and all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(w - 1))


def test_split_and_assemble_roundtrip():
    cs = open(os.path.join(GOLDEN_DIR, "g4_300x200_rgb16_53_rct_tile128.j2k"), "rb").read()
    hdr, parts = sharding.split_tileparts(cs)
    assert len(parts) == 6 and all(p[:2] == b"\xff\x90" for p in parts)
    assert [int.from_bytes(p[4:6], "big") for p in parts] == list(range(6))
    assert sharding.assemble(hdr, parts) == cs


def _worker(rank, world, port, path, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cs = open(path, "rb").read()
        hdr, parts = sharding.split_tileparts(cs)
        first, count = sharding.partition_tiles(len(parts), world)[rank]
        bufs = None
        for frame in range(2):  # second pass reuses the receive buffers
            local = torch.frombuffer(bytearray(b"".join(parts[first:first + count])), dtype=torch.uint8)
            got, bufs = sharding.gather_tileparts(local, rank, world, bufs)
            if rank == 0:
                out = sharding.assemble(hdr, [bytes(t.numpy().tobytes()) for t in got])
                q.put(out == cs)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_over_gloo(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    path = os.path.join(GOLDEN_DIR, "g4_300x200_rgb16_53_rct_tile128.j2k")
    procs = [ctx.Process(target=_worker, args=(r, world, port, path, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) and q.get(timeout=5)


@pytest.mark.gpu
def test_gather_tileparts_over_rccl_world_1():
    """The nccl (= RCCL) branch of the exchange on the GPU box: a one-rank process group, device tensors.  (The N > 1
    transfers need an N-GPU node, which only the driver has; their code path is the gloo-tested one.)"""
    import torch
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cs = open(os.path.join(GOLDEN_DIR, "g4_300x200_rgb16_53_rct_tile128.j2k"), "rb").read()
        hdr, parts = sharding.split_tileparts(cs)
        local = torch.frombuffer(bytearray(b"".join(parts)), dtype=torch.uint8).cuda()
        # the pieces a multi-rank gather is made of, on one rank: the all-gather of the lengths and a device-to-device payload
        n = torch.tensor([local.numel()], dtype=torch.int64, device="cuda")
        lens = [torch.zeros_like(n)]
        dist.all_gather(lens, n)
        assert int(lens[0].item()) == local.numel()
        got, _ = sharding.gather_tileparts(local, 0, 1)
        assert sharding.assemble(hdr, [bytes(got[0].cpu().numpy().tobytes())]) == cs
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _exchange_worker(rank, world, port, path, q):
    """bench.py's exchange as four ranks run it: three encode threads per rank hand frames over out of order, with fewer
    staging slots than frames in flight, different payloads per frame; rank 0 must see every frame once, in order, whole."""
    import threading
    import time
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cs = open(path, "rb").read()
        hdr, parts = sharding.split_tileparts(cs)
        first, count = sharding.partition_tiles(len(parts), world)[rank]
        mine = b"".join(parts[first:first + count])
        nframes, nthreads = 9, 3
        seen = []

        def payload_of(frame):  # frame f carries the rank's tile-parts followed by f marker bytes: lengths differ per frame
            return mine + bytes([frame]) * frame

        def stage(payload, slot_buf):  # the staging copy: a tensor of its own that outlives the caller's buffer
            return torch.frombuffer(bytearray(payload), dtype=torch.uint8).clone(), slot_buf

        def on_frame(frame, got):
            body = [bytes(t.numpy().tobytes()) for t in got]
            ok = all(b.endswith(bytes([frame]) * frame) for b in body)
            stripped = [b[:len(b) - frame] for b in body]
            seen.append((frame, ok and sharding.assemble(hdr, stripped) == cs))

        ex = sharding.Exchange(rank, world, 2, stage, on_frame)

        def encode_thread(k):
            for f in range(k, nframes, nthreads):
                time.sleep(0.002 * ((f * 7 + rank * 3) % 5))  # frames finish in a different order on every rank
                ex.submit(f, payload_of(f))
        for rnd in range(2):  # a second run after reset(): buffers and the frame counter are reused
            ex.reset()
            ths = [threading.Thread(target=encode_thread, args=(k,)) for k in range(nthreads)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            ex.drain(nframes)
        ex.close()
        if rank == 0:
            q.put([f for f, _ in seen] == list(range(nframes)) * 2 and all(ok for _, ok in seen))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_exchange_thread_over_gloo_world_4():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    path = os.path.join(GOLDEN_DIR, "g4_300x200_rgb16_53_rct_tile128.j2k")
    procs = [ctx.Process(target=_exchange_worker, args=(r, 4, port, path, q)) for r in range(4)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5)


def test_exchange_surfaces_a_failing_stage_instead_of_hanging():
    """a stage() that raises (allocation failure on the staging tensor) must not leave the gather thread waiting
    for that frame for ever: submit() re-raises, the slot goes back, drain() and later submits raise the same error."""
    import threading

    import torch

    calls = []

    def stage(payload, buf):
        calls.append(payload)
        if payload == 1:
            raise MemoryError("staging tensor")
        return torch.tensor([payload], dtype=torch.uint8), buf

    ex = sharding.Exchange(0, 1, 2, stage)
    ex.submit(0, 0)
    with pytest.raises(MemoryError):
        ex.submit(1, 1)
    done = []
    t = threading.Thread(target=lambda: done.append(pytest.raises(MemoryError, ex.drain, 2)), daemon=True)
    t.start()
    t.join(10)
    assert not t.is_alive() and done, "drain() still waits for the frame whose staging failed"
    with pytest.raises(MemoryError):
        ex.submit(2, 2)
    ex.close()

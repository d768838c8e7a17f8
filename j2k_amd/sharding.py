"""Tile sharding across ranks and the one exchange step of the path.

JPEG 2000 tiles are independent (own DWT, own code-blocks, own tile-part), so an image -- or a batch
of frames -- shards with no mid-pipeline exchange: rank r encodes a contiguous block of tiles in
raster order and the tile-parts are concatenated in tile-index order behind the main header.  The
only collective is a variable-length gather of the compressed tile-parts on rank 0, which owns the
output sink: (1) all-gather of the byte counts, (2) point-to-point payload transfers into rank 0
(over RCCL these ride the 7 direct xGMI links of rank 0 concurrently; many-to-one, no ring).
The same code runs over gloo on CPU tensors (tests) and over nccl(=RCCL) on device tensors (bench).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def partition_tiles(num_tiles: int, world: int) -> list[tuple[int, int]]:
    """Contiguous blocks of tiles per rank, raster order: [(first, count)] with sizes differing by <= 1."""
    base, rem = divmod(num_tiles, world)
    out, first = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((first, n))
        first += n
    return out


def gather_tileparts(local: torch.Tensor, rank: int, world: int, recv_bufs: list | None = None):
    """Gather the ranks' tile-part byte strings (1-D uint8 tensors, any device) on rank 0.

    Returns (list of per-rank tensors in rank order, recv_bufs) on rank 0 and (None, None)
    elsewhere.  recv_bufs can be passed back in to reuse the receive buffers across frames."""
    if world == 1:
        return [local], recv_bufs
    n = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    lens = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(lens, n)
    # One grouped point-to-point batch: over RCCL the world-1 receives of rank 0 become a single launch
    # whose transfers ride rank 0's direct xGMI links side by side (separate irecv calls would run one
    # after the other, each at the rate of a single link).
    if rank != 0:
        for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, local, 0)]):
            q.wait()
        if local.is_cuda:  # the payload buffer belongs to the encoder and is reused by its next call
            torch.cuda.current_stream(local.device).synchronize()
        return None, None
    sizes = torch.cat(lens).tolist()
    if recv_bufs is None:
        recv_bufs = [None] * world
    ops = []
    for r in range(1, world):
        if recv_bufs[r] is None or recv_bufs[r].numel() < sizes[r]:
            recv_bufs[r] = torch.empty(int(sizes[r] * 1.1) + 4096, dtype=torch.uint8, device=local.device)
        ops.append(dist.P2POp(dist.irecv, recv_bufs[r][:sizes[r]], r))
    for q in dist.batch_isend_irecv(ops):
        q.wait()
    if local.is_cuda:
        torch.cuda.current_stream(local.device).synchronize()
    return [local] + [recv_bufs[r][:sizes[r]] for r in range(1, world)], recv_bufs


class Exchange:
    """The exchange step off the encode threads (bench.py's multi-rank modes, and what a host driving N ranks would do):
    a frame's tile-part leaves the encoder's buffer -- which the handle's next frame overwrites -- for a staging buffer,
    and ONE thread per rank gathers the frames on rank 0 strictly in frame order -- the same order on every rank, as a
    collective needs -- while the encode threads go on with the next frames.

    stage(payload, slot_buffer_or_None) -> (staged tensor, slot buffer): copies `payload` (whatever submit() was handed)
    into a tensor the gather may keep until it has run; on_frame(frame, parts): rank 0 only, called in frame order with the
    ranks' tile-parts (tensors valid until the next frame of the same rank arrives)."""

    def __init__(self, rank: int, world: int, depth: int, stage, on_frame=None, setup=None):
        import threading
        self.rank, self.world, self.stage, self.on_frame, self.setup = rank, world, stage, on_frame, setup
        self.cv = threading.Condition()
        self.ready = {}             # frame -> (slot, staged tensor)
        self.next = 0               # next frame to exchange
        self.slots = [None] * depth  # staging buffers (allocated by stage() on first use)
        self.avail = list(range(depth))
        self.recv = None
        self.error = None
        self.stop = False
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def submit(self, frame: int, payload):
        """Called by an encode thread when frame `frame` of this rank is ready.  A frame is admitted only inside the window
        [next, next + depth): the frame the gather is waiting for always finds a slot, however far the other encode threads
        have run ahead (slots handed out first come, first served would deadlock there)."""
        with self.cv:
            self.cv.wait_for(lambda: (frame < self.next + len(self.slots) and self.avail) or self.error)
            if self.error:
                raise self.error
            slot = self.avail.pop()
        try:
            staged, self.slots[slot] = self.stage(payload, self.slots[slot])
        except BaseException as ex:
            # the gather thread would wait for this frame for ever (and with it drain() and every other submit(), and the
            # other ranks in the collective): the failure becomes the exchange's, the slot goes back
            with self.cv:
                self.avail.append(slot)
                if self.error is None:
                    self.error = ex
                self.stop = True
                self.cv.notify_all()
            raise
        with self.cv:
            self.ready[frame] = (slot, staged)
            self.cv.notify_all()

    def _run(self):
        try:
            if self.setup:
                self.setup()
            while True:
                with self.cv:
                    self.cv.wait_for(lambda: self.stop or self.next in self.ready)
                    if self.stop and self.next not in self.ready:
                        return
                    slot, staged = self.ready.pop(self.next)
                parts, self.recv = gather_tileparts(staged, self.rank, self.world, self.recv)
                if self.on_frame and self.rank == 0:
                    self.on_frame(self.next, parts)
                with self.cv:
                    self.next += 1
                    self.avail.append(slot)
                    self.cv.notify_all()
        except BaseException as ex:  # surfaces in submit() / drain()
            with self.cv:
                self.error = ex
                self.cv.notify_all()

    def reset(self):
        with self.cv:
            self.next = 0

    def drain(self, count: int):
        """Returns when `count` frames have been exchanged (rank 0 holds every tile-part of them)."""
        with self.cv:
            self.cv.wait_for(lambda: self.next >= count or self.error)
            if self.error:
                raise self.error

    def close(self):
        with self.cv:
            self.stop = True
            self.cv.notify_all()
        self.thread.join()


def assemble(main_header, parts: list[bytes]) -> bytes:
    """Main header + tile-parts in tile order + EOC.  `main_header` is either the header bytes or the
    job's api.Params: then the file wrapper (JP2 boxes, if asked for) goes in front as well, its jp2c box
    length taken from the assembled codestream."""
    if isinstance(main_header, (bytes, bytearray)):
        return bytes(main_header) + b"".join(parts) + b"\xff\xd9"
    from . import api
    cs = api.main_header(main_header) + b"".join(parts) + b"\xff\xd9"
    return api.file_header(main_header, len(cs)) + cs


def split_tileparts(codestream: bytes) -> tuple[bytes, list[bytes]]:
    """Inverse of assemble (used by tests): (main header, [tile-part bytes]) via the SOT Psot fields."""
    pos = codestream.index(b"\xff\x90")
    header, parts = codestream[:pos], []
    while codestream[pos:pos + 2] == b"\xff\x90":
        psot = int.from_bytes(codestream[pos + 6:pos + 10], "big")
        parts.append(codestream[pos:pos + psot])
        pos += psot
    assert codestream[pos:] == b"\xff\xd9"
    return header, parts

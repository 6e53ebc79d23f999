"""Multi-GPU layout of the VFO chain: one process per GPU under torch.distributed.

The path shards trivially: sub VFOs share only read-only inputs and keep private
state, so each rank owns a static block of the sub VFOs of every main VFO, the 2-3 main VFOs are
replicated on every rank, and the ONLY exchange is the raw IQ frame, broadcast from the ingest rank
once per frame (RCCL over xGMI when the backend is "nccl").  Results leave each GPU by its own
D2H copy; there is no reduce / gather on the data path.

The compute engine is injected (``make_engine(topology_shard) -> obj with process(frame)``), so
the same orchestration runs the HIP library on GPUs and -- in the CPU test-suite, backend "gloo" --
the CPU oracle.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .topology import Topology, shard


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def force_collectives() -> bool:
    """SDRX_FORCE_COLLECTIVES=1: run the collective code path even in a 1-rank job -- on a 1-GPU box this is the
    only way the RCCL ("nccl") branch, its communication stream and its events ever execute (the broadcast then
    has one participant; transport over xGMI is of course not exercised)."""
    return os.environ.get("SDRX_FORCE_COLLECTIVES") == "1"


def init_process_group(backend: str | None = None, device: torch.device | None = None) -> tuple[int, int]:
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun's env)."""
    rank, world, _ = env_world()
    if (world > 1 or force_collectives()) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


class FrameBroadcast:
    """Double-buffered broadcast of raw frames from `src_rank`: `frames_per_batch` frames (each
    2*n_complex float32) travel in ONE collective, so its launch and the stream synchronisation
    around it are paid once per batch, not once per frame (4 frames = 1 s of signal = 12.3 MB at
    1.536 MS/s; the latency a real receiver pays for batching is the ingest's, not the GPUs').

    Two ways to use it:
      * ``buf = bc(frames)``: broadcast now, on the current stream (simple, serial);
      * ``bc.submit(frames)`` ... ``buf = bc.result()``: the broadcast of the NEXT batch runs on a
        communication stream of its own while the current batch is processed.  ``result()`` makes
        the current (compute) stream wait for the pending broadcast.  ``submit()`` records ONE event
        on the current stream that (a) orders the communication stream behind whatever produced
        `frames` and (b) marks the buffer handed out by the previous ``result()`` as consumed --
        so call it AFTER the work that reads that buffer has been enqueued on the current stream
        (the Receiver must run on that stream: ``Receiver.set_stream``).  ``consumed()`` does (b)
        alone for callers that want to release a buffer earlier."""

    def __init__(self, n_complex: int, device: torch.device, src_rank: int = 0, frames_per_batch: int = 1):
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.src = src_rank
        self.frames_per_batch = int(frames_per_batch)
        self.frame_floats = 2 * n_complex
        self.buf = [torch.empty(self.frames_per_batch * 2 * n_complex, dtype=torch.float32, device=device) for _ in range(2)]
        self.k = 0
        self.cuda = device.type == "cuda"
        self.single = self.world == 1 and not (force_collectives() and dist.is_initialized())  # nothing to exchange
        self.comm = torch.cuda.Stream(device) if (self.cuda and not self.single) else None
        self._pending = None            # (buffer, work handle or None)
        self._free = [None, None]       # per buffer: event after which it may be overwritten
        self._last = None               # index of the buffer handed out by the last result()

    def frame(self, batch: torch.Tensor, j: int) -> torch.Tensor:
        """Frame j of a batch returned by result() / __call__."""
        return batch[j * self.frame_floats:(j + 1) * self.frame_floats]

    def __call__(self, frame: torch.Tensor | None) -> torch.Tensor:
        """`frame`: the new batch on the source rank (ignored elsewhere).  Returns this
        rank's copy, valid until the call after next."""
        if self.single:
            return frame
        b = self.buf[self.k & 1]
        self.k += 1
        if self.rank == self.src:
            b.copy_(frame, non_blocking=True)
        dist.broadcast(b, src=self.src)
        return b

    # -- overlapped form ------------------------------------------------------------------------
    def submit(self, frame: torch.Tensor | None) -> None:
        if self.single:
            self._pending = (frame, None)
            return
        i = self.k & 1
        self.k += 1
        b = self.buf[i]
        if self.comm is None:  # CPU tensors (gloo in the test-suite): nothing to overlap with
            if self.rank == self.src:
                b.copy_(frame)
            self._pending = (b, dist.broadcast(b, src=self.src, async_op=True), i)
            return
        # one event on the caller's stream: everything enqueued so far -- the producer of `frame`
        # and the consumer of the previously returned buffer -- is ahead of it
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        if self._last is not None:
            self._free[self._last] = ev
        with torch.cuda.stream(self.comm):
            if self.rank == self.src:
                self.comm.wait_event(ev)
            if self._free[i] is not None and self._free[i] is not ev:
                self.comm.wait_event(self._free[i])  # the batch that last used this buffer has been consumed
            if self.rank == self.src:
                b.copy_(frame, non_blocking=True)
            work = dist.broadcast(b, src=self.src, async_op=True)
        self._pending = (b, work, i)

    def result(self) -> torch.Tensor:
        assert self._pending is not None, "result() without submit()"
        if self.single:
            b, _ = self._pending
            self._pending = None
            return b
        b, work, i = self._pending
        self._pending = None
        work.wait()  # NCCL: the CURRENT stream waits for the collective; gloo: the host does
        self._last = i
        return b

    def consumed(self) -> None:
        """Marks, in the current stream's order, the point after which the buffer returned by the last
        result() may be overwritten (submit() does the same)."""
        if self.single or not self.cuda or self._last is None:
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._free[self._last] = ev


class ShardedReceiver:
    """This rank's share of a profile: shard(topology, rank, world) on an injected engine."""

    def __init__(self, topo: Topology, make_engine, device: torch.device | None = None, src_rank: int = 0):
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.full = topo
        self.topo = shard(topo, self.rank, self.world)
        # more ranks than sub VFOs: this rank holds nothing and only takes part in the broadcast
        self.engine = make_engine(self.topo) if self.topo.vfos else None
        self.device = device or torch.device("cpu")
        self.bcast = FrameBroadcast(topo.frame, self.device, src_rank)

    def process(self, frame: torch.Tensor | None):
        """One frame: broadcast the raw IQ, run the local shard.  Returns the local frame tensor."""
        local = self.bcast(frame)
        if self.engine is not None:
            self.engine.process(local)
        return local

    def leaf_topics(self) -> list[str]:
        return [self.topo.vfos[i].topic for i in self.topo.leaves_in_publish_order()]

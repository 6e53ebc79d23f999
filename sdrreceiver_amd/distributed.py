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


def init_process_group(backend: str | None = None, device: torch.device | None = None) -> tuple[int, int]:
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun's env)."""
    rank, world, _ = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


class FrameBroadcast:
    """Double-buffered broadcast of the raw frame (2*n_complex float32) from `src_rank`.

    Two ways to use it:
      * ``buf = bc(frame)``: broadcast now, on the current stream (simple, serial);
      * ``bc.submit(frame)`` ... ``buf = bc.result()`` ... ``bc.consumed()``: the broadcast of the
        NEXT frame runs on a communication stream of its own while the current frame is being
        processed.  ``result()`` makes the current (compute) stream wait for the pending broadcast,
        ``consumed()`` marks, in compute-stream order, the point after which the buffer returned by
        the previous ``result()`` may be overwritten."""

    def __init__(self, n_complex: int, device: torch.device, src_rank: int = 0):
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.src = src_rank
        self.buf = [torch.empty(2 * n_complex, dtype=torch.float32, device=device) for _ in range(2)]
        self.k = 0
        self.cuda = device.type == "cuda"
        self.comm = torch.cuda.Stream(device) if (self.cuda and self.world > 1) else None
        self._pending = None            # (buffer, work handle or None)
        self._free = [None, None]       # per buffer: event after which it may be overwritten
        self._last = None               # index of the buffer handed out by the last result()

    def __call__(self, frame: torch.Tensor | None) -> torch.Tensor:
        """`frame`: the new raw frame on the source rank (ignored elsewhere).  Returns this
        rank's copy, valid until the call after next."""
        if self.world == 1:
            return frame
        b = self.buf[self.k & 1]
        self.k += 1
        if self.rank == self.src:
            b.copy_(frame, non_blocking=True)
        dist.broadcast(b, src=self.src)
        return b

    # -- overlapped form ------------------------------------------------------------------------
    def submit(self, frame: torch.Tensor | None) -> None:
        if self.world == 1:
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
        with torch.cuda.stream(self.comm):
            if self._free[i] is not None:
                self.comm.wait_event(self._free[i])  # the frame that last used this buffer has been consumed
            if self.rank == self.src:
                b.copy_(frame, non_blocking=True)
            work = dist.broadcast(b, src=self.src, async_op=True)
        self._pending = (b, work, i)

    def result(self) -> torch.Tensor:
        assert self._pending is not None, "result() without submit()"
        if self.world == 1:
            b, _ = self._pending
            self._pending = None
            return b
        b, work, i = self._pending
        self._pending = None
        work.wait()  # NCCL: the CURRENT stream waits for the collective; gloo: the host does
        self._last = i
        return b

    def consumed(self) -> None:
        if self.world == 1 or not self.cuda or self._last is None:
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
        self.engine = make_engine(self.topo)
        self.device = device or torch.device("cpu")
        self.bcast = FrameBroadcast(topo.frame, self.device, src_rank)

    def process(self, frame: torch.Tensor | None):
        """One frame: broadcast the raw IQ, run the local shard.  Returns the local frame tensor."""
        local = self.bcast(frame)
        self.engine.process(local)
        return local

    def leaf_topics(self) -> list[str]:
        return [self.topo.vfos[i].topic for i in self.topo.leaves_in_publish_order()]

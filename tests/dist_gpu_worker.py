"""Worker of tests/test_distributed_gpu.py: one rank of a 2-rank job on ONE GPU (backend gloo --
RCCL refuses two ranks on the same device).  Rank 0 owns the raw frames and broadcasts them; every
rank runs its shard of the sdr_25E tree on the HIP library and checks it, bit for bit, against the
CPU oracle fed with independently regenerated frames."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import binding as ob  # noqa: E402
from sdrreceiver_amd import distributed as D, synth, topology as tp  # noqa: E402
from sdrreceiver_amd.receiver import Receiver  # noqa: E402


class Engine:
    def __init__(self, topo):
        self.rx = Receiver.from_topology(topo, device=0, exact=True)

    def process(self, frame: torch.Tensor):
        torch.cuda.current_stream().synchronize()  # the broadcast result is complete
        self.rx.process_device(frame.data_ptr(), frame.numel() // 2)


def main():
    rank, world = D.init_process_group("gloo")
    dev = torch.device("cuda", 0)
    # SDRX_DIST_TREE=config5-<n>: BASELINE config 5's tree shape (config-3 rule, n sub VFOs in total,
    # sharded over the ranks); default: the sdr_25E profile
    tree = os.environ.get("SDRX_DIST_TREE", "25e")
    full = tp.config5(int(tree.split("-")[1])) if tree.startswith("config5-") else tp.profile_25e()
    threads = max(1, len(os.sched_getaffinity(0)) // world)
    sr = D.ShardedReceiver(full, Engine, device=dev)
    nodes, roots = ob.build_tree("port", sr.topo)
    lcg = synth.Lcg(1)
    for f in range(3):
        iq = synth.lcg_frame(full.frame, lcg)  # every rank can regenerate the frame for its checker ...
        src = torch.from_numpy(iq).to(dev) if rank == 0 else None  # ... but only rank 0 feeds the GPUs
        sr.process(src)
        ob.process_roots(roots, iq, threads=threads)
        rx = sr.engine.rx
        for i in sr.topo.leaves_in_publish_order():
            assert np.array_equal(rx.output(i), nodes[i].usb()), (rank, f, sr.topo.vfos[i].topic)
    # the overlapped broadcast (the form bench.py uses): frame k+1 travels on a communication stream
    # while frame k is processed; 6 more frames through the same receiver, checked the same way
    bc = sr.bcast
    frames = [synth.lcg_frame(full.frame, lcg) for _ in range(6)]
    dev = [torch.from_numpy(f).to(dev_) if rank == 0 else None for f, dev_ in ((f, dev) for f in frames)]
    bc.submit(dev[0])
    for f in range(6):
        b = bc.result()
        sr.engine.process(b)
        bc.consumed()
        if f + 1 < 6:
            bc.submit(dev[f + 1])
        ob.process_roots(roots, frames[f], threads=threads)
        for i in sr.topo.leaves_in_publish_order():
            assert np.array_equal(sr.engine.rx.output(i), nodes[i].usb()), (rank, "overlap", f, sr.topo.vfos[i].topic)
    topics = sr.leaf_topics()
    gathered = [None] * world
    dist.all_gather_object(gathered, topics)
    if rank == 0:
        allt = sorted(t for g in gathered for t in g)
        want = sorted(full.vfos[i].topic for i in full.leaves_in_publish_order())
        assert allt == want, (allt, want)
        print(f"OK {len(allt)} leaves over {world} ranks: {[len(g) for g in gathered]}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""The N>1 path on CPU: 2 processes, gloo backend, the distributed orchestration of
sdrreceiver_amd/distributed.py with the CPU oracle injected as the compute engine.  Checks the
frame broadcast and that the union of the shards equals the unsharded result bit for bit."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _OracleEngine:
    def __init__(self, topo):
        from oracle import binding as ob
        self.topo = topo
        self.nodes, self.roots = ob.build_tree("port", topo)
        self.ob = ob

    def process(self, frame):
        self.ob.process_roots(self.roots, frame.cpu().numpy())

    def outputs(self):
        return {self.topo.vfos[i].topic: self.nodes[i].usb() for i in self.topo.leaves_in_publish_order()}


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from sdrreceiver_amd import distributed as D, synth, topology as tp
    D.init_process_group("gloo")
    topo = tp.config3(16)
    sr = D.ShardedReceiver(topo, _OracleEngine)
    lcg = synth.Lcg(1)
    res = []
    for f in range(3):
        frame = torch.from_numpy(synth.lcg_frame(topo.frame, lcg)) if rank == 0 else None
        local = sr.process(frame)
        res.append((float(local.double().sum()), {k: v.copy() for k, v in sr.engine.outputs().items()}))
    # the overlapped form the bench uses: the broadcast of frame k+1 is in flight while frame k is processed
    bc = D.FrameBroadcast(topo.frame, torch.device("cpu"))
    lcg2 = synth.Lcg(7)
    sums = []
    bc.submit(torch.from_numpy(synth.lcg_frame(topo.frame, lcg2)) if rank == 0 else None)
    for f in range(4):
        b = bc.result()
        sums.append(float(b.double().sum()))  # "processing" frame f ...
        bc.consumed()
        bc.submit(torch.from_numpy(synth.lcg_frame(topo.frame, lcg2)) if rank == 0 else None)  # ... while f+1 travels
    bc.result()
    want = synth.Lcg(7)
    assert sums == [float(synth.lcg_frame(topo.frame, want).astype(np.float64).sum()) for _ in range(4)], sums
    # several frames per collective (what bench.py does at N > 1): one broadcast carries 4 frames
    bc4 = D.FrameBroadcast(topo.frame, torch.device("cpu"), frames_per_batch=4)
    lcg4 = synth.Lcg(11)
    batches = [np.concatenate([synth.lcg_frame(topo.frame, lcg4) for _ in range(4)]) for _ in range(3)]
    bc4.submit(torch.from_numpy(batches[0]) if rank == 0 else None)
    for k in range(3):
        b = bc4.result()
        for j in range(4):
            fr = bc4.frame(b, j)
            assert fr.numel() == 2 * topo.frame
            assert float(fr.double().sum()) == float(batches[k][j * 2 * topo.frame:(j + 1) * 2 * topo.frame].astype(np.float64).sum()), (k, j)
        if k + 1 < 3:
            bc4.submit(torch.from_numpy(batches[k + 1]) if rank == 0 else None)
    # more ranks than sub VFOs: config 1 has ONE sub, which the partition gives to the last rank; rank 0
    # holds nothing, still takes part in the broadcast, and no rank turns the main VFO into a leaf
    c1 = tp.config1()
    sr1 = D.ShardedReceiver(c1, _OracleEngine)
    lcg1 = synth.Lcg(2)
    f1 = synth.lcg_frame(c1.frame, lcg1)
    sr1.process(torch.from_numpy(f1) if rank == 0 else None)
    if rank == 0:
        assert sr1.engine is None and sr1.leaf_topics() == []
    else:
        from oracle import binding as ob
        nodes, roots = ob.build_tree("port", c1)
        ob.process_roots(roots, f1)
        assert sr1.leaf_topics() == ["VFO01"] and np.array_equal(sr1.engine.outputs()["VFO01"], nodes[1].usb())
    # weak-scaling bookkeeping the bench uses: totals are sums over ranks
    t = torch.tensor([float(sr.topo.vfo_samples_per_frame())], dtype=torch.float64)
    dist.all_reduce(t)
    q.put((rank, res, float(t.item()), sr.leaf_topics()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process():
    sys.path.insert(0, ROOT)
    from oracle import binding as ob
    from sdrreceiver_amd import synth, topology as tp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    topo = tp.config3(16)
    nodes, roots = ob.build_tree("port", topo)
    lcg = synth.Lcg(1)
    topics0, topics1 = got[0][3], got[1][3]
    assert len(topics0) == len(topics1) == 8 and not set(topics0) & set(topics1)
    # replicated mains: each rank counts them, the sub VFOs are counted once
    assert got[0][2] == got[1][2] == topo.vfo_samples_per_frame() + 2 * topo.frame
    for f in range(3):
        frame = synth.lcg_frame(topo.frame, lcg)
        ob.process_roots(roots, frame)
        ref = {topo.vfos[i].topic: nodes[i].usb() for i in topo.leaves_in_publish_order()}
        assert got[0][1][f][0] == got[1][1][f][0] == float(frame.astype(np.float64).sum())  # same frame everywhere
        merged = {**got[0][1][f][1], **got[1][1][f][1]}
        assert set(merged) == set(ref)
        for k in ref:
            assert np.array_equal(merged[k], ref[k]), (f, k)

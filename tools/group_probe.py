"""What the single-process multi-device orchestration costs: BASELINE config 3 as ONE context vs the same tree
as a group of m contexts (sdrx_group_*) that all sit on the one GPU -- the m shards then do the same work
in total, so the difference is the per-frame fan-out (one event, m waits, m-1 device copies of the 3 MB
frame) plus what smaller launches lose.  Frames stay on the device (no payload copies)."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdrreceiver_amd import synth, topology as tp
from sdrreceiver_amd.receiver import Receiver, Group

n_subs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
topo = tp.config3(n_subs)
fr = torch.from_numpy(synth.lcg_frame(topo.frame, synth.Lcg(1))).cuda()
torch.cuda.synchronize()
N = 40 if n_subs <= 2048 else 10


def measure(step, sync):
    for _ in range(max(20, int(0.05 / (1e-4 * n_subs / 1024)))):
        step()
    sync()
    reps = []
    for _ in range(9):
        t0 = time.perf_counter()
        for _ in range(N):
            step()
        sync()
        reps.append((time.perf_counter() - t0) / N * 1e3)
    return statistics.median(reps)


rx = Receiver.from_topology(topo, device=0)
print(f"one context, {n_subs} subs: {measure(lambda: rx.process_device(fr.data_ptr(), topo.frame), rx.sync):.4f} ms per frame", flush=True)
rx.close()
for m in (1, 2, 4, 8):
    g = Group.from_topology(topo, [0] * m)
    print(f"group of {m} context(s) on one GPU: {measure(lambda: g.process_device(fr.data_ptr(), topo.frame), g.sync):.4f} ms per frame", flush=True)
    g.close()

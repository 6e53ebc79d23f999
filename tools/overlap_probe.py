"""Upper bound of cross-frame overlap: two independent receivers (same tree) on two streams, frames
alternating between them, against one receiver doing the same number of frames on one stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdrreceiver_amd import synth, topology as tp
from sdrreceiver_amd.receiver import Receiver

topo = tp.config3(1024)
fr = torch.from_numpy(synth.lcg_frame(topo.frame, synth.Lcg(1))).cuda()
rxs, streams = [], []
for i in range(3):
    rx = Receiver.from_topology(topo, device=0, exact=True)
    st = torch.cuda.Stream()
    rx.set_stream(st.cuda_stream)
    rxs.append(rx); streams.append(st)
N = 60
for n_ctx in (1, 2, 3):
    for _ in range(6):
        for i in range(n_ctx):
            rxs[i].process_device(fr.data_ptr(), topo.frame)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(N):
        rxs[k % n_ctx].process_device(fr.data_ptr(), topo.frame)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n_ctx} stream(s): {dt / N * 1e3:.4f} ms per frame", flush=True)

"""Per-class launch-time probe (run on the GPU box): sub launches made only of d=5 subs (384 k ->
12 k) or only of d=2 subs (192 k -> 48 k), at several counts, to separate the per-item cost of
k_mix_decimate / k_usb_demod from the per-launch fixed cost."""
import copy, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdrreceiver_amd import synth, topology as tp
from sdrreceiver_amd.receiver import Receiver


def only(topo, keep):
    t = copy.deepcopy(topo)
    t.vfos = [v for v in t.vfos if v.parent < 0 or keep(v)]
    return t


def run(topo, steps=20, segments=0):
    rx = Receiver.from_topology(topo, device=0, exact=True, segments=segments)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st); rx.set_stream(st.cuda_stream)
    fr = torch.from_numpy(synth.lcg_frame(topo.frame, synth.Lcg(1))).cuda()
    for _ in range(5):
        rx.process_device(fr.data_ptr(), topo.frame)
    torch.cuda.synchronize()
    rx.enable_kernel_timing(True)
    for _ in range(steps):
        rx.process_device(fr.data_ptr(), topo.frame)
    torch.cuda.synchronize()
    kt = rx.kernel_times()
    rx.close()
    return {k: round(v["ms"] / v["launches"] * 1e3, 1) for k, v in kt.items()}


if __name__ == "__main__":
    seg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    for n in (256, 512, 1024, 2048, 4096, 8192):
        full = tp.config3(2 * n)
        for name, keep in (("d5", lambda v: v.decimate_count == 5), ("d2", lambda v: v.decimate_count == 2)):
            r = run(only(full, keep), segments=seg)
            print(json.dumps({"class": name, "n": n, "us": r}), flush=True)
    for n in (1024, 2048, 4096):
        print(json.dumps({"class": "config3", "n": n, "us": run(tp.config3(n), segments=seg)}), flush=True)

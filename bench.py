#!/usr/bin/env python3
"""bench.py -- throughput of the per-VFO IQ chain on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config3|flat|config2|config4|config5|10k|64k|256k]
                    [--fast] [--no-cpu] [--configs1]

A "step" is one pass of the hot path over one raw IQ frame (250 ms of signal: 384 000 cf32 at
1.536 MS/s), already resident in HBM, through every VFO of the workload.  Default workload =
BASELINE.json config 3: the two sdr_25E main VFOs with 512 sub VFOs each (1 024 sub VFOs; main0
subs 384 k -> 12 k, main1 subs 192 k -> 48 k, every 2nd with the 47-tap 10 kHz low-pass).

Multi-GPU (one process per GPU under torch.distributed.run): the sub VFOs shard across ranks
with no data-path collective except the one the path really has -- the raw frame is broadcast
from rank 0 over RCCL every step.  Weak scaling: every rank runs the full single-GPU workload
(N x 1 024 sub VFOs in total).

Prints ONE JSON line (rank 0).  `value` = IQ MSamples/s ingested, summed over every VFO chain of
every rank (the unit that scales with the number of VFOs); `raw_iq_msps` and `vfos_at_realtime`
are the other two readings of BASELINE.json's metric.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def make_topology(name, world):
    from sdrreceiver_amd import topology as tp
    if name == "config3":
        return tp.config3(1024 * world), "BASELINE config 3: 2 main VFOs (1.536 MS/s -> 384 k / 192 k) + 1024 sub VFOs per GPU"
    if name == "flat":
        return tp.config3_flat(1024 * world), "flat variant: 1024 leaf VFOs per GPU, each 1.536 MS/s -> 48 kHz (d=5)"
    if name == "config2":
        return tp.config2(), "BASELINE config 2: 32 sub VFOs across the 2 sdr_25E main VFOs"
    if name == "config4":
        return tp.config4(256 * world), "BASELINE config 4: 1.92 MS/s, 3 mains, 256 late-decimate subs per GPU, 10 kHz LPF"
    if name == "10k":
        return tp.config3(10240 * world), "north-star target: 10 240 sub VFOs per GPU under the 2 sdr_25E mains"
    if name == "config5":
        return tp.config5(65536), "BASELINE config 5: 65 536 sub VFOs in total, sharded over the GPUs (strong scaling), raw frame broadcast"
    if name == "256k":
        return tp.config3(262144 * world), "memory-scale check: 262 144 sub VFOs per GPU under the 2 sdr_25E mains (~70 GB of HBM)"
    if name == "64k":
        return tp.config3(65536 * world), "BASELINE config 5's tree on ONE GPU: 65 536 sub VFOs under the 2 sdr_25E mains"
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(workload):
    """The reference's CPU path on a bounded sample of the same workload, on this box's host
    cores.  Prefers the real reference build (oracle/_ref, kind "reference", one thread -- how
    the reference actually runs); falls back to the plain-C restatement (kind
    "port").  The all-cores OpenMP figure of the port is reported next to it."""
    import numpy as np
    from oracle import binding as ob
    from sdrreceiver_amd import synth, topology as tp
    if workload == "flat":
        topo, frames = tp.config3_flat(16), 8
    elif workload == "config4":
        topo, frames = tp.config4(96), 12
    elif workload == "config2":
        topo, frames = tp.config2(), 40
    else:
        topo, frames = tp.config3(256), 12
    sample = f"{topo.name}: {len(topo.vfos)} VFOs x {frames} frames of {topo.frame} cf32 (LCG input)"
    iq = synth.lcg_frame(topo.frame, synth.Lcg(1))
    out = {}

    def run(kind, threads):
        nodes, roots = ob.build_tree(kind, topo)
        ob.process_roots(roots, iq, frames=1, threads=threads)  # warm caches / first touch
        t0 = time.perf_counter()
        ob.process_roots(roots, iq, frames=frames, threads=threads)
        dt = time.perf_counter() - t0
        for r in roots:
            r.free()
        return frames * topo.vfo_samples_per_frame() / dt / 1e6

    kind = "port"
    if ob.have_reference():
        try:
            ob.load("reference")
            kind = "reference"
        except OSError:
            kind = "port"
    v1 = run(kind, 1)
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    ncores = min(ncores, 64)  # OpenMP over sub VFOs stops scaling long before that on this workload
    out = {"value": round(v1, 3), "unit": "MSamples/s", "cores": 1, "kind": kind, "sample": sample}
    try:
        out["port_1thread"] = round(run("port", 1), 3) if kind != "port" else out["value"]
        out["port_all_cores"] = {"value": round(run("port", ncores), 3), "cores": ncores}
    except Exception as e:  # pragma: no cover
        out["port_error"] = str(e)
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
        out["cpu"] = model[0] if model else "unknown"
    except OSError:
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="config3")
    ap.add_argument("--fast", action="store_true", help="FMA arithmetic (within 1e-6 of the reference) instead of bit-exact")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--segments", type=int, default=0)
    ap.add_argument("--configs1", action="store_true",
                    help="also time BASELINE configs[1] (32 sub VFOs) and report it as a side reading (off by default: the "
                         "profiled default command must launch the kernels of ONE workload only)")
    args = ap.parse_args()

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libsdrx has no CPU fallback)")
    # SDRX_BENCH_SHARE_GPU=1 (validation on a 1-GPU box only): all ranks on device 0, gloo instead of
    # RCCL (which refuses two ranks on one device).  The numbers of such a run mean nothing.
    share = os.environ.get("SDRX_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    from sdrreceiver_amd import distributed as D, synth, topology as tp
    from sdrreceiver_amd.receiver import Receiver
    dist = None
    if world > 1:
        import torch.distributed as dist
        D.init_process_group("gloo" if share else "nccl", device=torch.device("cuda", local))

    full, descr = make_topology(args.workload, world)
    topo = tp.shard(full, rank, world)
    rx = Receiver.from_topology(topo, device=local, exact=not args.fast, segments=args.segments)
    stream = torch.cuda.Stream()  # a real (non-null) stream shared by torch, RCCL ordering and our kernels
    torch.cuda.set_stream(stream)
    rx.set_stream(stream.cuda_stream)
    st = rx.stats()

    # the raw frame lives in HBM; rank 0 owns the source, the others receive it by broadcast
    frame_np = synth.lcg_frame(topo.frame, synth.Lcg(1))
    src = torch.from_numpy(frame_np).cuda() if rank == 0 else None
    bcast = D.FrameBroadcast(topo.frame, torch.device("cuda", local), src_rank=0)  # RCCL over xGMI: the only exchange

    # The broadcast of frame k+1 (RCCL, its own stream) overlaps the processing of frame k; at N = 1
    # submit/result hand the resident frame straight through.
    overlap = True
    try:
        bcast.submit(src)
        bcast.result()
        bcast.consumed()
        bcast.submit(src)
    except Exception as e:  # fall back to the serial broadcast rather than lose the run
        overlap = False
        if rank == 0:
            print(f"bench: overlapped broadcast unavailable ({type(e).__name__}: {e}); broadcasting in line", file=sys.stderr)

    def step(k):
        if not overlap:
            b = bcast(src)
            rx.process_device(b.data_ptr(), topo.frame)
            return
        b = bcast.result()
        rx.process_device(b.data_ptr(), topo.frame)
        bcast.consumed()
        bcast.submit(src)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([float(st["vfo_samples_per_frame"]), float(st["algorithmic_bytes_per_frame"]),
                            float(st["n_leaves"])], dtype=torch.float64, device="cuda")
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        vfo_samples, alg_bytes, n_leaves = (float(x) for x in tot.tolist())
    else:
        vfo_samples, alg_bytes, n_leaves = float(st["vfo_samples_per_frame"]), float(st["algorithmic_bytes_per_frame"]), float(st["n_leaves"])

    # second, separate pass with HIP events around every kernel launch (on the launch stream):
    # the dominant kernel's average duration for the roofline object
    rx.enable_kernel_timing(True)
    for k in range(min(args.steps, 20)):
        step(k)
    barrier()
    kt = rx.kernel_times()
    rx.enable_kernel_timing(False)
    # third: with the payload D2H copy + publish callbacks in the loop (never `value`)
    barrier()
    t1 = time.perf_counter()
    for k in range(min(args.steps, 10)):
        step(k)
        rx.fetch()
    barrier()
    dt_d2h = (time.perf_counter() - t1) / min(args.steps, 10)
    # fourth: the same without the (Python, ctypes) callbacks -- what a C / C++ host sees: kernels + payload D2H
    rx.set_publish(False)
    barrier()
    t1 = time.perf_counter()
    for k in range(min(args.steps, 10)):
        step(k)
        rx.fetch()
    barrier()
    dt_d2h_nocb = (time.perf_counter() - t1) / min(args.steps, 10)
    rx.set_publish(True)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        frame_seconds = topo.frame / topo.fs
        value = args.steps * vfo_samples / dt / 1e6
        kernels = {}
        dom, dom_ms = None, -1.0
        frame_kernel_ms = 0.0
        for name, r in kt.items():
            avg = r["ms"] / r["launches"]
            per_frame = r["ms"] / min(args.steps, 20)
            frame_kernel_ms += per_frame
            kernels[name] = {"avg_ms": round(avg, 5), "launches_per_frame": r["launches"] // min(args.steps, 20),
                             "kernel_bytes_per_launch": r["alg_bytes"] // r["launches"],
                             "GBps": round(r["alg_bytes"] / r["launches"] / (avg * 1e-3) / 1e9, 1)}
            if per_frame > dom_ms:
                dom, dom_ms = name, per_frame
        # Roofline of the dominant kernel.  Algorithmic bytes (8*n_in + W_out per
        # VFO per frame) for the VFOs that kernel processes in one launch / its average duration.
        d = kt[dom]
        dom_bytes = d["alg_bytes"] / d["launches"]
        dom_avg_s = d["ms"] / d["launches"] * 1e-3
        achieved = dom_bytes / dom_avg_s / 1e9
        traffic, traffic_src = None, None
        try:  # HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/profile.sh)
            pm = json.load(open(os.path.join(ROOT, "profiles", "current_pmc.json")))
            if pm.get("workload") == args.workload and pm.get("exact") == (not args.fast):
                k = pm["kernels"].get(dom)
                if k:
                    traffic, traffic_src = int(k["hbm_bytes_per_launch"]), pm.get("source")
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "IQ MSamples/s ingested, summed over VFO chains (1.536 MS/s -> 48/12 kHz USB chain)",
            "value": round(value, 2), "unit": "MSamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong" if args.workload == "config5" else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": descr, "name": full.name, "vfos_total": int(len(full.vfos)), "sub_vfos_per_gpu": int(st["n_leaves"]),
                       "frame_cf32": topo.frame, "fs": topo.fs, "arithmetic": "fast-fma" if args.fast else "exact (bit-identical to -O2 reference)",
                       "parallelism": f"vfo-shard x{world}, raw frame RCCL broadcast" if world > 1 else "single GPU"},
            "raw_iq_msps": round(args.steps * topo.frame / dt / 1e6, 2),
            "vfos_at_realtime": int(n_leaves * frame_seconds / (dt / args.steps)),
            "realtime_factor": round(frame_seconds / (dt / args.steps), 1),
            "algorithmic_GBps_whole_frame": round(args.steps * alg_bytes / dt / 1e9, 1),
            "ms_per_step_with_payload_d2h_and_callbacks": round(dt_d2h * 1e3, 4),
            "ms_per_step_with_payload_d2h": round(dt_d2h_nocb * 1e3, 4),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "bytes_per_launch": int(dom_bytes), "avg_launch_ms": round(dom_avg_s * 1e3, 5),
                         "frame_kernel_ms": round(frame_kernel_ms, 5),
                         "frame_frac": round(alg_bytes / world / (frame_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "kernels": kernels,
        }
        if world == 1 and args.configs1:
            # BASELINE.json configs[1] (32 sub VFOs), the same way, as a side reading: a latency-bound
            # plumbing case on this hardware (three ~10 us launches per frame)
            try:
                t2 = tp.config2()
                rx2 = Receiver.from_topology(t2, device=local, exact=not args.fast)
                rx2.set_stream(stream.cuda_stream)
                st2 = rx2.stats()
                for _ in range(args.warmup):
                    rx2.process_device(src.data_ptr(), t2.frame)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    rx2.process_device(src.data_ptr(), t2.frame)
                torch.cuda.synchronize()
                d2 = (time.perf_counter() - t0) / args.steps
                rx2.close()
                out["configs1_32_sub_vfos"] = {"ms_per_step": round(d2 * 1e3, 4), "value": round(st2["vfo_samples_per_frame"] / d2 / 1e6, 2),
                                               "unit": "MSamples/s", "realtime_factor": round(frame_seconds / d2, 1),
                                               "algorithmic_GBps_whole_frame": round(st2["algorithmic_bytes_per_frame"] / d2 / 1e9, 1)}
            except Exception as e:
                out["configs1_32_sub_vfos"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu:
            try:
                out["cpu_baseline"] = cpu_baseline(args.workload)
            except Exception as e:  # the bench line must still come out
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out))
    if overlap:
        bcast.result()  # drain the broadcast that is still in flight before the process group goes away
    barrier()
    rx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

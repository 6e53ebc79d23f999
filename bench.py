#!/usr/bin/env python3
"""bench.py -- throughput of the per-VFO IQ chain on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reps R]
                    [--workload config3|flat|flat10k|config2|config4|config5|10k|64k|256k|512k]
                    [--fast] [--full] [--budget-s S] [--no-cpu] [--no-abi] [--configs1] [--batch B]

A "step" is one pass of the hot path over one raw IQ frame (250 ms of signal: 384 000 cf32 at
1.536 MS/s), already resident in HBM, through every VFO of the workload.

Workload.  BASELINE.json config 3 PER GPU -- the two sdr_25E main VFOs with 512 sub VFOs each (1 024 sub
VFOs; main0 subs 384 k -> 12 k, main1 subs 192 k -> 48 k, every 2nd with the 47-tap 10 kHz low-pass).
N = 1 is the BENCH line; at N > 1 (the SCALE lines) the tree has N x 1 024 sub VFOs, sharded over the N
GPUs (mains replicated), raw frames broadcast from rank 0 over RCCL: the path partitions, per-GPU work
is fixed, `scaling` is "weak", and `value` at N is directly comparable with N x the N = 1 value.
BASELINE.json config 5 -- the same tree with 65 536 sub VFOs IN TOTAL (strong scaling: 65 536 / N per
GPU) -- is measured in the same run and reported as the side object `config5_strong`.  `--workload`
overrides the default (with `--workload config5` the roles swap: side object `weak_config3`).

Timing.  W untimed warm-up steps (plus enough extra to reach ~50 ms of GPU time: the clock ramps),
then the timed region -- EXACTLY K steps between barrier + torch.cuda.synchronize() on both sides,
max over ranks -- is repeated R times (default 11); `ms_per_step` and `value` are the MEDIAN
repetition, min / max are in `ms_per_step_min/max`.

Prints ONE JSON line of at most 8 KB (rank 0): the contract keys, `roofline`, `cpu_baseline`, `verified` and one compact
row per side workload.  Everything else that is measured (the counters' VALU sub-objects, the instruction mix, the
through-the-ABI details, the prose) goes to `bench_full.json` beside this file (and to `gpurun_out/` where that exists) and
to stderr.  `value` = IQ MSamples/s ingested, summed over every VFO chain of every rank (the unit that scales with the
number of VFOs); `raw_iq_msps` and `vfos_at_realtime` are the other two readings of BASELINE.json's metric.

Time.  The default command is held to ~30 s of wall time (`--budget-s`): the clock is warmed on a throw-away receiver, so
the oracle that verifies the timed launch sequence replays a few hundred frames, not a thousand; the side workloads run in
order of importance while the budget lasts (a side that did not fit says so in its row); `--full` lifts the budget and
adds the Qt drop-in legs.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# option "exact" of include/sdrx.h (True == 1, False == 0: the same keys)
ARITH = {1: "exact (bit-identical to -O2 reference)",
         0: "tolerance (<= 1e-5 of max|ref|, int16 within 1 LSB: NCO as rotations of its exact checkpoints, FMA mixer and filters)",
         2: "robust (<= 1e-5 of max|ref| whatever is out of band, int16 within 1 LSB: the table NCO exact, FMA mixer and filters)"}
N_SIMD = 256 * 4       # 256 CUs x 4 SIMDs


def make_topology(name, world):
    from sdrreceiver_amd import topology as tp
    if name == "config3":
        return tp.config3(1024 * world), "BASELINE config 3: 2 main VFOs (1.536 MS/s -> 384 k / 192 k) + 1024 sub VFOs per GPU"
    if name == "flat":
        return tp.config3_flat(1024 * world), "flat variant: 1024 leaf VFOs per GPU, each 1.536 MS/s -> 48 kHz (d=5)"
    if name == "flat10k":
        return tp.config3_flat(10240 * world), ("the north-star sentence read literally: 10 240 leaf VFOs per GPU, EACH consuming the 1.536 MS/s "
                                                "stream (1.536 MS/s -> 48 kHz, d=5, no parent)")
    if name == "config2":
        return tp.config2(), "BASELINE config 2: 32 sub VFOs across the 2 sdr_25E main VFOs"
    if name == "config4":
        return tp.config4(256 * world), "BASELINE config 4: 1.92 MS/s, 3 mains, 256 late-decimate subs per GPU, 10 kHz LPF"
    if name == "10k":
        return tp.config3(10240 * world), "north-star target: 10 240 sub VFOs per GPU under the 2 sdr_25E mains"
    if name == "config5":
        return tp.config5(65536), (f"BASELINE config 5: 65 536 sub VFOs in total under the 2 sdr_25E mains, sharded over {world} GPU(s) "
                                   "(strong scaling), raw frames broadcast from rank 0")
    if name == "256k":
        return tp.config3(262144 * world), "memory-scale check: 262 144 sub VFOs per GPU under the 2 sdr_25E mains (~70 GB of HBM)"
    if name == "512k":
        return tp.config3(524288 * world), "memory-scale check: 524 288 sub VFOs per GPU under the 2 sdr_25E mains (~140 GB of HBM)"
    if name == "768k":
        return tp.config3(786432 * world), "memory-scale check: 786 432 sub VFOs per GPU under the 2 sdr_25E mains (~220 GB of HBM)"
    if name == "64k":
        return tp.config3(65536 * world), "BASELINE config 5's tree on ONE GPU: 65 536 sub VFOs under the 2 sdr_25E mains"
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(workload):
    """The reference's CPU path on a bounded sample of the same workload, on this box's host
    cores.  Prefers the real reference build (oracle/_ref, kind "reference", one thread -- how
    the reference actually runs); falls back to the plain-C restatement (kind
    "port").  The all-cores OpenMP figure of the port is reported next to it."""
    from oracle import binding as ob
    from sdrreceiver_amd import synth, topology as tp
    if workload in ("flat", "flat10k"):
        topo, frames = tp.config3_flat(8), 8
    elif workload == "config4":
        topo, frames = tp.config4(48), 12
    elif workload == "config2":
        topo, frames = tp.config2(), 20
    else:
        topo, frames = tp.config3(128), 12   # ~115 M VFO-samples: ~2.2 s of the reference's one thread (+ the port's two readings; measured 6 s in all)
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


class Verifier:
    """Checks WHAT WAS TIMED: the launch sequence of the timed region (sdrx_process_device back to back on the torch stream:
    one k_mix_levels launch per step, level l on frame k - l, + the leaf tail of the frame that left the last level) at the
    benchmarked size.  After every timed repetition -- outside the region -- `checkpoint()` fetches the payloads of the frame
    processed last (sdrx_fetch: completes what is queued, copies the payloads out) and keeps those of a seeded sample of
    leaves (topology.sample_leaves: first / last leaf of every shard block of every main + >= 64 random ones), then queues
    one untimed frame so that the next region starts from the same steady pipeline state.  `finish()` runs the plain-C
    oracle (oracle/vfo_oracle.c, the checker -- never the thing measured) over EVERY frame this receiver was handed since
    its creation, warm-up and timed frames alike, and compares at every checkpoint: exact arithmetic bit for bit, --fast
    within 1 LSB int16 / 1e-5 of max|ref| on the final cf32 stream (north_star's tolerance).  sdrj.cpp:288-294."""

    def __init__(self, job, n_random=64, seed=20261002):
        from sdrreceiver_amd import topology as tp
        self.job = job
        self.sample = tp.sample_leaves(job.topo, n_random, seed)
        self.sub, self.remap = tp.subset(job.topo, self.sample)
        self.points = []   # (frames handed over so far, {leaf: payload}, {leaf: stream} | None)

    def checkpoint(self, with_streams=False):
        j = self.job
        j.rx.set_publish(False)  # (a ctypes callback per leaf costs microseconds: the payloads are read through sdrx_get_output)
        j.rx.fetch()
        j.rx.set_publish(True)
        pay = {i: j.rx.output(i) for i in self.sample}
        st = {i: j.rx.stream(i, missing_ok=True) for i in self.sample} if with_streams else None
        self.points.append((len(j.hist), pay, st))
        j.step(len(j.hist))     # refill the frame pipeline: untimed, and part of the history like every other frame
        j.realign(len(j.hist))

    def finish(self, exact, rel_tol=1e-5):
        import numpy as np
        from oracle import binding as ob
        j = self.job
        t0 = time.perf_counter()
        threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        # (two short parallel regions per frame: 256 threads on the 2 x 64-core host took 30x as long as 32; at N > 1 every rank runs
        # its own oracle at the same time: the ranks share the host's cores)
        threads = max(1, min(threads // max(1, int(os.environ.get("WORLD_SIZE", "1"))), 32))
        nodes, roots = ob.build_tree("port", self.sub)
        done, bad, worst, checked = 0, [], 0.0, 0
        try:
            for upto, pay, st in self.points:
                k = done
                while k < upto:  # runs of the same frame go through the oracle in one call
                    e = k
                    while e < upto and j.hist[e] == j.hist[k]:
                        e += 1
                    ob.process_roots(roots, j.frames_np[j.hist[k]], frames=e - k, threads=threads)
                    k = e
                done = upto
                for i in self.sample:
                    ref = nodes[self.remap[i]]
                    want = ref.usb() if j.topo.vfos[i].demod_usb else ref.iq()
                    got = pay[i]
                    if exact:
                        ok = np.array_equal(got, want)
                    else:
                        ok = got.shape == want.shape and int(np.abs(got.astype(np.int32) - want.astype(np.int32)).max()) <= 1
                    if ok and st is not None and st[i] is not None:
                        rs = ref.stream()
                        if exact:
                            ok = np.array_equal(st[i].view(np.uint64), rs.view(np.uint64))
                        else:
                            err = float(np.abs(st[i] - rs).max()) / max(float(np.abs(rs).max()), 1e-30)
                            worst = max(worst, err)
                            ok = err <= rel_tol
                    checked += 1
                    if not ok:
                        bad.append((upto, i))
        finally:
            for r in roots:
                r.free()
        out = {"ok": not bad, "leaves": len(self.sample), "checkpoints": len(self.points), "frames": done,
               "compared": "int16 payloads at every checkpoint (after each timed repetition) + final cf32 streams at the last one; "
                           + ("bit for bit" if exact else f"+-1 LSB / {rel_tol:g} of max|ref|"),
               "against": "oracle/vfo_oracle.c (plain-C restatement, pinned to the compiled reference) over every frame of the run",
               "leaf_checks": checked, "oracle_s": round(time.perf_counter() - t0, 1), "oracle_threads": threads}
        if not exact:
            out["worst_stream_rel_err"] = float(f"{worst:.3g}")
        if bad:
            out["mismatches"] = [{"after_frames": a, "vfo": b} for a, b in bad[:8]]
        return out


def pmc_for(workload, exact):
    """profiles/current_pmc.json (tools/profile.sh + tools/pmc_summary.py): per-launch counter means of
    the committed PMC passes, if they were taken on this workload and arithmetic."""
    suffix = {1: "", 0: "_tolerance", 2: "_robust"}[int(exact)]
    for name in (f"pmc_{workload}{suffix}.json", "current_pmc.json"):  # per-workload summaries next to the default one
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        if pm.get("workload") == workload and int(pm.get("exact", -1)) == int(exact):
            return pm
    return None


def build_id():
    try:
        from sdrreceiver_amd import _lib
        return _lib.lib().sdrx_build_id().decode()
    except Exception:
        return "unknown"


def demanded_valu_per_launch(topo, exact=True):
    """VALU wave-instructions the arithmetic demands of ONE steady-state launch of the mix/decimate items (one frame's worth
    of every VFO).  EXACT -- the reference's operations in its order: per 1024-sample chunk NCO replay 16 x 7 + mix 16 x 3,
    stage 0 8 x 11 + 16 halo moves, stage 1 4 x 11 + 16, an 11-instruction dot product per 64 outputs of every deeper stage
    (tools/inst_mix.py checks these counts against the compiled ISA); per 960 / 1008-sample chunk of a fused late
    decimation NCO + mix and 3 x Nd x 2 for the decimating low-pass.  TOLERANCE arithmetic: table entry and mixer 16 x (2 + 2),
    a half-band output 3 pair sums + 1 product + 3 FMAs = 7, a low-pass tap one FMA.  ROBUST arithmetic: the exact table replay
    16 x 7 + the mixer 16 x 2, filters as in the tolerance arithmetic.  Addressing, loop control, warm-up: not demanded."""
    nco_mix, hb, mac = (16 * 10, 11, 2) if int(exact) == 1 else (16 * 9, 7, 1) if int(exact) == 2 else (16 * 4, 7, 1)
    total = 0
    for v in topo.vfos:
        n, d = v.samples_per_buffer, v.decimate_count
        if v.demod_usb and v.late_decimate in (5, 6) and d == 0 and v.parent >= 0:
            chunk, taps = (960, 49) if v.late_decimate == 5 else (1008, 73)
            total += -(-n // chunk) * (nco_mix + 3 * taps * mac)
            continue
        per = nco_mix
        if d >= 1:
            per += 8 * hb + 16
        if d >= 2:
            per += 4 * hb + 16
        for s_ in range(2, d):
            per += hb * (1024 >> (s_ + 1)) / 64.0  # (a stage with fewer than 64 outputs per chunk: a fraction of a wave-instruction)
        total += -(-n // 1024) * per
    return int(total)


def roofline_object(dom, d, kt_steps, frame_kernel_ms, alg_bytes, world, pm, mix_chunks, demanded=None):
    """The dominant launch against the bound that binds it.
    `bound` / `achieved` / `peak` / `unit` / `frac` are those of the LIMITER the counters name (a fraction by construction):
    "hbm" -- counter bytes (what really crossed the L2 <-> fabric boundary) per second against 8 TB/s -- or "valu" -- VALU
    issue slots carrying an instruction against what a saturated probe of the same instruction mix sustains.
    SURVEY.md 8d's contract figure (ALGORITHMIC bytes -- 8 n_in for EVERY sibling although they share one parent stream
    through L2 -- over the launch duration) is `algorithmic_GBps` / `algorithmic_over_hbm_peak`: it may exceed 1 and is not
    called a fraction."""
    dom_bytes = d["alg_bytes"] / d["launches"]
    dom_avg_s = d["ms"] / d["launches"] * 1e-3
    alg = dom_bytes / dom_avg_s / 1e9
    r = {"bound": None, "kernel": dom, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None, "traffic_source": None,
         "algorithmic_GBps": round(alg, 1), "algorithmic_over_hbm_peak": round(alg / HBM_PEAK_GBS, 4),
         "bytes_per_launch": int(dom_bytes), "avg_launch_ms": round(dom_avg_s * 1e3, 5),
         "frame_kernel_ms": round(frame_kernel_ms, 5),
         "frame_frac": round(alg_bytes / world / (frame_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    k = pm["kernels"].get(dom) if pm else None
    if not k:
        r["limited_by"] = "unknown here: no committed PMC passes for this workload (tools/profile.sh)"
        return r
    r["traffic_source"] = pm.get("source")
    r["pmc_git_sha"] = pm.get("git_sha")
    # the counter passes name the library they measured (sdrx_build_id, a hash of the sources and flags): a mismatch
    # means the committed counters are from another build than the one timed here
    r["pmc_build_id"] = pm.get("build_id")
    r["pmc_matches_build"] = (pm.get("build_id") == build_id()) if pm.get("build_id") else None
    hb = None
    if "hbm_bytes_per_launch" in k:
        r["traffic"] = int(k["hbm_bytes_per_launch"])
        hb = min(1.0, r["traffic"] / dom_avg_s / 1e9 / HBM_PEAK_GBS)
        r["frac_hbm_unique"] = round(hb, 4)
        r["traffic_over_algorithmic"] = round(r["traffic"] / dom_bytes, 3)
    if "l2_hit_rate" in k:
        r["l2_hit_rate"] = k["l2_hit_rate"]
    v = k.get("valu")
    busy = None
    if v:
        # tools/pmc_summary.py: busy quad-cycles = SQ_INSTS_VALU - SQ_ACTIVE_INST_VALU2 (the SIMD pairs plain fp32 ops),
        # cycles = SQ_BUSY_CYCLES / 32 of the SAME pass, normalised by what a saturated probe of the kernel's own
        # instruction mix reads under the same counters (tools/valu_calib.hip, profiles/valu_calibration.json)
        valu = {kk: v[kk] for kk in ("valu_busy", "valu_busy_raw", "calibration_probe", "probe_reads_raw", "dual_issued_frac", "valu_insts",
                                     "cycles", "clock_GHz", "pass_dur_us", "wave_cycles_split", "lds_inst_busy")
                if kk in v}
        busy = min(1.0, valu.get("valu_busy", valu.get("valu_busy_raw", 0.0)))
        valu["busy"] = busy
        valu["issued_insts_per_launch"] = v["valu_insts"]
        if mix_chunks and dom.startswith("k_mix"):
            valu["issued_insts_per_chunk"] = round(v["valu_insts"] / mix_chunks, 1)
        if demanded and dom.startswith("k_mix"):
            # the part of the launch's VALU capacity spent on instructions the reference's arithmetic demands
            valu["demanded_insts_per_launch"] = int(demanded)
            if mix_chunks:
                valu["demanded_insts_per_chunk"] = round(demanded / mix_chunks, 1)
            valu["useful_frac"] = round(min(busy, 4.0 * demanded / (1024 * v["cycles"]) / valu.get("probe_reads_raw", 1.0)), 4)
        if "inst_mix" in pm and dom.startswith("k_mix"):
            valu["inst_mix_per_chunk"] = pm["inst_mix"]
        r["valu"] = valu
    if busy is not None and (hb is None or busy > hb):
        slots = v["valu_insts"] * (1.0 - v.get("dual_issued_frac", 0.0) / 2.0)  # issue slots carrying an instruction
        r["bound"], r["unit"] = "valu", "G VALU issue slots/s"
        r["achieved"] = round(slots / (v["pass_dur_us"] * 1e3), 1)
        # the peak from the hardware and the calibration alone: 1024 SIMDs x one VALU issue slot per 4 cycles at the clock the
        # counters of that pass saw, x what a saturated probe of the same instruction mix reads under the same counters
        # (tools/valu_calib.hip).  frac = achieved / peak is then a CHECK of the calibrated busy reading (`valu.busy`, derived
        # from the cycle counter instead of the duration), not an identity: tests/test_bench_contract.py holds the two together.
        r["peak"] = round(N_SIMD * v["clock_GHz"] / 4.0 * v.get("probe_reads_raw", 1.0), 1)
        r["frac"] = round(min(1.0, r["achieved"] / r["peak"]), 4)
        r["limited_by"] = (f"VALU issue: {busy:.0%} of the launch's issue slots carry a VALU instruction (calibrated) -- not HBM: "
                           + (f"real traffic is {hb:.0%} of the 8 TB/s peak, siblings share the parent's stream through L2" if hb is not None
                              else "no traffic counters"))
    elif hb is not None:
        r["bound"], r["unit"], r["peak"] = "hbm", "GB/s", HBM_PEAK_GBS
        r["achieved"] = round(r["traffic"] / dom_avg_s / 1e9, 1)
        r["frac"] = round(hb, 4)
        r["limited_by"] = f"HBM: {hb:.0%} of the 8 TB/s peak in real traffic (counter bytes)" + (f"; VALU issue {busy:.0%}" if busy is not None else "")
    r["limiter"] = r["bound"]
    return r


ROOFLINE_LINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_GBps",
                      "algorithmic_over_hbm_peak", "frac_hbm_unique", "bytes_per_launch", "avg_launch_ms", "frame_kernel_ms", "frame_frac",
                      "l2_hit_rate", "pmc_build_id", "pmc_matches_build")
SIDE_KEYS = ("north_star_10k", "flat_10k", "flat_1024", "config4_256", "fast_config3", "fast_10k", "fast_config4", "robust_config3",
             "robust_10k", "robust_config4", "config5_64k_one_gpu", "config5_strong", "weak_config3", "configs1_32_sub_vfos")
LINE_LIMIT = 8192  # bytes: what the driver's parser is known to take (the 22.7 KB line of round 5 was not parsed)


def side_row(o):
    """One compact row of a side workload for the printed line (its whole object is in bench_full.json)."""
    if not isinstance(o, dict):
        return o
    if "ms_per_step" not in o:
        return {k: o[k] for k in ("skipped", "error") if k in o}
    r = {"ms_per_step": o["ms_per_step"]}
    for k in ("sub_vfos", "sub_vfos_total", "realtime_factor", "frame_frac", "scaling", "value"):
        if k in o:
            r[k] = o[k]
    if "arithmetic" in o:
        r["arithmetic"] = o["arithmetic"].split(" ", 1)[0]
    rf = o.get("roofline") or {}
    if rf:
        r["bound"], r["frac"] = rf.get("bound"), rf.get("frac")
    if isinstance(o.get("verified"), dict):
        r["verified_ok"] = o["verified"].get("ok")
        if "worst_stream_rel_err" in o["verified"]:
            r["worst_rel_err"] = o["verified"]["worst_stream_rel_err"]
    return r


def compact_line(full, limit=LINE_LIMIT):
    """The printed line: the contract keys + roofline + cpu_baseline + verified + one row per side workload, <= `limit`
    bytes.  Optional parts are dropped (least important first) should it ever come out longer."""
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_min", "ms_per_step_max",
                                 "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "build_id", "repetitions", "raw_iq_msps",
                                 "vfos_at_realtime", "realtime_factor", "algorithmic_GBps_whole_frame", "frame_frac_of_hbm_roofline",
                                 "ms_per_step_through_abi", "ms_per_step_with_payload_d2h", "rccl_world", "peer_ok", "wall_s") if k in full}
    c = full.get("config", {})
    clip = lambda x: (x[:297] + "...") if isinstance(x, str) and len(x) > 300 else x   # noqa: E731  (prose never decides the line's size)
    line["config"] = {k: clip(c[k]) for k in ("workload", "name", "arithmetic", "vfos_total", "sub_vfos_per_gpu", "parallelism") if k in c}
    if "roofline" in full:
        line["roofline"] = {k: full["roofline"][k] for k in ROOFLINE_LINE_KEYS if k in full["roofline"]}
        v = full["roofline"].get("valu")
        if v:
            line["roofline"]["valu_busy"] = v.get("busy")
            if "useful_frac" in v:
                line["roofline"]["valu_useful_frac"] = v["useful_frac"]
    if "cpu_baseline" in full:
        line["cpu_baseline"] = {k: clip(v) for k, v in full["cpu_baseline"].items()}
    if isinstance(full.get("verified"), dict):
        line["verified"] = {k: full["verified"][k] for k in ("ok", "leaves", "frames", "checkpoints", "error", "ranks_with_mismatches", "worst_stream_rel_err")
                            if k in full["verified"]}
    if "kernels" in full:
        line["kernels"] = {k: {"avg_ms": v["avg_ms"]} for k, v in full["kernels"].items()}
    a = full.get("through_abi")
    if a:
        t = {k: a[k] for k in ("sync_pageable_ms", "pipelined_pageable_ms", "pipelined_u8_ms", "u8_dc_sync_ms", "u8_dc_pipelined_ms", "u8_dc_kernels_ms")
             if k in a}
        if isinstance(a.get("c_host"), dict):
            t["c_host"] = {k: v for k, v in a["c_host"].items() if k.endswith("_ms") or k == "error"}
        if isinstance(a.get("qt_adapter"), dict):
            t["qt_adapter"] = {k: v for k, v in a["qt_adapter"].items() if k.endswith("_ms")}
        line["through_abi"] = t
    for k in SIDE_KEYS:
        if k in full:
            line[k] = side_row(full[k])
    line["full"] = "bench_full.json"
    for drop in ("kernels", "through_abi", "raw_iq_msps", "repetitions", "algorithmic_GBps_whole_frame") + tuple(reversed(SIDE_KEYS)):
        if len(json.dumps(line)) <= limit:
            break
        line.pop(drop, None)
    return line


def write_full(full):
    """Everything that was measured: bench_full.json beside this file (+ gpurun_out/, which travels back from the GPU box)
    and stderr.  Never stdout: the driver reads ONE line there."""
    text = json.dumps(full, indent=1)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_full.json"), "w") as f:
                    f.write(text + "\n")
            except OSError:
                pass
    print("bench_full: " + json.dumps(full), file=sys.stderr)


def main():
    t_start = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reps", type=int, default=0, help="repetitions of the timed K-step region (median reported; default 11, 25 with --full)")
    ap.add_argument("--workload", default=None)
    ap.add_argument("--fast", action="store_true",
                    help="the tolerance arithmetic (option exact = 0: within 1e-5 of the reference, north_star's bar) instead of bit-exact")
    ap.add_argument("--arith", type=int, default=None, choices=[0, 1, 2],
                    help="option \"exact\" of the library: 1 exact (default), 0 tolerance (= --fast), 2 robust (exact NCO, FMA mixer and filters)")
    ap.add_argument("--full", action="store_true",
                    help="no time budget: every side workload, 25 repetitions, a larger oracle sample, the Qt drop-in legs")
    ap.add_argument("--budget-s", type=float, default=32.0,
                    help="wall-time budget of the default command: side workloads are started only while it lasts")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-abi", action="store_true", help="skip the through-the-ABI (host buffers, PCIe both ways) leg")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the oracle check of the timed launch sequence (`verified` in the line; a profiled command skips it: "
                         "the checkpoints add fetches and single-level launches to the kernel statistics)")
    ap.add_argument("--no-side", action="store_true",
                    help="skip the side readings of the N = 1 line (north-star 10 240 subs, flat, config 4, ...): a profiled "
                         "command must launch the kernels of ONE workload only")
    ap.add_argument("--no-clock-warmup", action="store_true", help="no throw-away receiver spinning the clock up before the timed regions")
    ap.add_argument("--pipeline", action="store_true", help="leaf tail on a second stream beside the next frame's levels (A/B switch; slower)")
    ap.add_argument("--no-fuse", action="store_true", help="one k_mix_decimate launch per tree level instead of k_mix_levels (A/B switch)")
    ap.add_argument("--no-frame-pipeline", action="store_true", help="k_mix_levels, but every frame runs through all its levels at once (A/B switch)")
    ap.add_argument("--segments", type=int, default=0)
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="an option of sdrreceiver_amd.receiver.Receiver for every receiver of the run (A/B switches: fuse_demod=0, fuse_late=0, ...)")
    ap.add_argument("--batch", type=int, default=0, help="frames per broadcast at N > 1 (default 4)")
    ap.add_argument("--configs1", action="store_true",
                    help="also time BASELINE configs[1] (32 sub VFOs) and report it as a side reading (off by default: the "
                         "profiled default command must launch the kernels of ONE workload only)")
    args = ap.parse_args()
    arith = args.arith if args.arith is not None else (0 if args.fast else 1)  # option "exact" of the line's own workload
    args.fast = arith != 1                                                      # (the side readings in the other arithmetics ride along with the exact line only)
    n_reps = args.reps or (25 if args.full else 11)
    user_options = {k: int(v) for k, v in (o.split("=", 1) for o in args.option)}
    budget = float("inf") if args.full else args.budget_s
    legs = {}  # wall seconds per leg of the run (bench_full.json): where the command's time goes

    def left():
        return budget - (time.perf_counter() - t_start)

    class leg:
        def __init__(self, name):
            self.name = name

        def __enter__(self):
            self.t0 = time.perf_counter()

        def __exit__(self, *a):
            legs[self.name] = round(legs.get(self.name, 0.0) + time.perf_counter() - self.t0, 2)

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
    workload = args.workload or "config3"
    # SDRX_BENCH_SHARE_GPU=1 (validation on a 1-GPU box only): all ranks on device 0, gloo instead of
    # RCCL (which refuses two ranks on one device).  The numbers of such a run mean nothing.
    share = os.environ.get("SDRX_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    # a launcher that gives every rank ONE visible device (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES = its own GPU): that device is
    # ordinal 0 in this process, whatever LOCAL_RANK says -- and rank 0's device is not visible, so peer access cannot be asked here
    one_visible = (not share) and world > 1 and torch.cuda.device_count() == 1
    if one_visible:
        local = 0
    torch.cuda.set_device(local)
    from sdrreceiver_amd import distributed as D, synth, topology as tp
    from sdrreceiver_amd.receiver import Receiver
    dist = None
    use_dist = world > 1 or D.force_collectives()  # (SDRX_FORCE_COLLECTIVES=1: the RCCL path with one rank, on a 1-GPU box)
    if use_dist:
        import torch.distributed as dist
        D.init_process_group("gloo" if share else "nccl", device=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    stream = torch.cuda.Stream()  # a real (non-null) stream shared by torch, RCCL ordering and our kernels
    torch.cuda.set_stream(stream)
    batch = args.batch or (4 if use_dist else 1)
    legs["start_up"] = round(time.perf_counter() - t_start, 2)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def allmax(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def allsum(vals):
        if not use_dist:
            return [float(v) for v in vals]
        t = torch.tensor([float(v) for v in vals], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(x) for x in t.tolist()]

    diag_world, diag_peer = None, None
    if use_dist:
        # so that the first run on several devices diagnoses itself: what the process group says, who reaches whom
        diag_world = dist.get_world_size()
        try:
            ok = 1 if (rank == 0 or share) else -1 if one_visible else int(torch.cuda.can_device_access_peer(local, 0))
        except Exception:
            ok = -1
        if os.environ.get("SDRX_BENCH_FAKE_NO_PEER") == "1" and rank == world - 1:
            ok = 0  # (tests/test_distributed_gpu.py: what the refusal below looks like, on a box where every rank reaches rank 0)
        t = torch.zeros(world, dtype=torch.int64, device="cuda")
        t[rank] = ok
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        diag_peer = [int(x) for x in t.tolist()]
        if (diag_world != args.gpus or 0 in diag_peer) and os.environ.get("SDRX_BENCH_ALLOW_NO_PEER") != "1":
            # a run whose ranks cannot reach rank 0's device directly (the broadcast would be staged through the host) or whose
            # process group is not the one asked for measures the wrong thing: say why and fail instead of printing a slow number
            if rank == 0:
                print(f"bench: refusing to measure: --gpus {args.gpus} but the process group has {diag_world} ranks; "
                      f"hipDeviceCanAccessPeer(rank's device -> rank 0's device) per rank = {diag_peer} (0 = no direct xGMI / PCIe peer "
                      "access: the raw-frame broadcast would go through host memory).  Check HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES, "
                      "the IOMMU / ACS settings and `rocm-smi --showtopo`; SDRX_BENCH_ALLOW_NO_PEER=1 measures anyway.", file=sys.stderr)
            dist.destroy_process_group()
            sys.exit(4)

    class ClockWarmer:
        """Spins the clock up before a timed region: ~50 ms of the same kernels on a throw-away receiver (config 3, 1 024 subs)
        that nothing verifies -- a cold 5 ms measurement reads 0.137 ms per frame where the warm one reads 0.110.
        Kept apart from the measured receivers so that the oracle, which replays EVERY frame a measured receiver was handed,
        replays a few hundred frames instead of a thousand."""

        def __init__(self):
            self.rx, self.per_step = None, None
            if args.no_clock_warmup:
                return
            t = tp.config3(1024)
            self.rx = Receiver.from_topology(t, device=local, exact=arith, **user_options)
            self.rx.set_stream(stream.cuda_stream)
            self.frame = t.frame
            self.src = torch.from_numpy(synth.lcg_frame(t.frame, synth.Lcg(7))).to(dev)

        def spin(self, seconds=0.05):
            if not self.rx:
                return
            if self.per_step is None:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(40):
                    self.rx.process_device(self.src.data_ptr(), self.frame)
                torch.cuda.synchronize()
                self.per_step = max((time.perf_counter() - t0) / 40, 2e-5)
            for _ in range(int(min(4000, seconds / self.per_step))):
                self.rx.process_device(self.src.data_ptr(), self.frame)

        def close(self):
            if self.rx:
                torch.cuda.synchronize()
                self.rx.close()
                self.rx = None

    with leg("clock_warmer"):
        warmer = ClockWarmer()

    class Job:
        """One workload on this rank's shard: the Receiver, the raw-frame source and the broadcast."""

        def __init__(self, name, exact=None, options=None):
            self.exact = arith if exact is None else int(exact)   # option "exact": 1 | 0 tolerance | 2 robust
            self.full, self.descr = make_topology(name, world)
            self.topo = tp.shard(self.full, rank, world)
            self.frame = self.full.frame
            self.rx = Receiver.from_topology(self.topo, device=local, exact=self.exact, segments=args.segments,
                                             pipeline=args.pipeline, fuse=not args.no_fuse,
                                             frame_pipeline=not args.no_frame_pipeline, **dict(user_options, **(options or {}))) if self.topo.vfos else None
            if self.rx:
                self.rx.set_stream(stream.cuda_stream)
            self.st = self.rx.stats() if self.rx else {"vfo_samples_per_frame": 0, "algorithmic_bytes_per_frame": 0, "n_leaves": 0,
                                                        "mix_chunks_per_frame": 0}
            # the raw frames live in HBM; rank 0 owns the source (`batch` different frames), the others
            # receive them by broadcast -- the ONLY exchange of the path (RCCL over xGMI)
            lcg = synth.Lcg(1)
            self.frames_np = [synth.lcg_frame(self.frame, lcg) for _ in range(batch)]
            self.src = torch.from_numpy(np.concatenate(self.frames_np)).to(dev) if rank == 0 else None
            self.bcast = D.FrameBroadcast(self.frame, dev, src_rank=0, frames_per_batch=batch)
            self.cur = None
            self.hist = []  # which of the `batch` source frames every step so far was handed (the verifier replays it)
            self.overlap = True
            try:  # the broadcast of batch k+1 (RCCL, its own stream) overlaps the processing of batch k
                self.bcast.submit(self.src)
            except Exception as e:  # fall back to the serial broadcast rather than lose the run
                self.overlap = False
                if rank == 0:
                    print(f"bench: overlapped broadcast unavailable ({type(e).__name__}: {e}); broadcasting in line", file=sys.stderr)

        def step(self, k, fetch=False):
            j = k % batch
            if j == 0:
                self.cur = self.bcast.result() if self.overlap else self.bcast(self.src)
            self.hist.append(j)
            if self.rx:
                self.rx.process_device(self.cur.data_ptr() + j * self.frame * 8, self.frame)
                if fetch:
                    self.rx.fetch()
            if j == batch - 1 and self.overlap:
                self.bcast.submit(self.src)  # one event: batch consumed, next one may travel

        def timed(self, steps, fetch=False):
            barrier()
            t0 = time.perf_counter()
            for k in range(steps):
                self.step(k, fetch)
            barrier()
            return allmax(time.perf_counter() - t0)

        def measure(self, steps, warmup, reps, verifier=None):
            for k in range(warmup):
                self.step(k)
            self.realign(warmup)
            warmer.spin()                   # the clock: ~50 ms of GPU time on the throw-away receiver, queued in front of the region
            if args.no_clock_warmup:        # (A/B: the round-5 way, warm-up frames through the measured receiver itself)
                dt = self.timed(steps)
                self.realign(steps)
                extra = int(min(2000, max(0, 0.05 / max(dt / steps, 1e-7) - steps - warmup)))
                for k in range(extra):
                    self.step(k)
                self.realign(extra)
            out = []
            for _ in range(reps):
                out.append(self.timed(steps))
                self.realign(steps)
                if verifier:
                    verifier.checkpoint()  # outside the timed region: payloads of the frame the region ended on
            return out

        def realign(self, steps_done):
            """Finish a partly used batch so that the next region starts at a batch boundary."""
            k = steps_done % batch
            while k % batch:
                self.step(k)
                k += 1

        def close(self):
            if self.overlap:
                self.bcast.result()  # drain the broadcast that is still in flight before the process group goes away
            barrier()
            if self.rx:
                self.rx.close()

    def kernel_pass(j, kt_steps):
        """A separate pass with HIP events around every kernel launch (on the launch's stream): per-kernel
        average durations, the dominant kernel and the frame's summed kernel time."""
        kt = {}
        if j.rx:
            j.rx.enable_kernel_timing(True)
            for k in range(kt_steps):
                j.step(k)
            j.realign(kt_steps)
            barrier()
            kt = j.rx.kernel_times()
            j.rx.enable_kernel_timing(False)
        else:
            barrier()
        kernels, dom, dom_ms, frame_kernel_ms = {}, None, -1.0, 0.0
        for name, r in kt.items():
            if r["launches"] * 2 < kt_steps:
                continue  # the one or two single-level launches that fill / drain the frame pipeline
            avg = r["ms"] / r["launches"]
            per_frame = r["ms"] / kt_steps
            frame_kernel_ms += per_frame
            kernels[name] = {"avg_ms": round(avg, 5), "launches_per_frame": int(round(r["launches"] / kt_steps)),
                             "kernel_bytes_per_launch": r["alg_bytes"] // r["launches"],
                             "GBps": round(r["alg_bytes"] / r["launches"] / (avg * 1e-3) / 1e9, 1)}
            if per_frame > dom_ms:
                dom, dom_ms = name, per_frame
        return kt, kernels, dom, frame_kernel_ms

    def side_reading(key, name, exact=None, need_s=3.0, options=None):
        """Another workload of BASELINE.json / SURVEY.md 8d on this GPU (or the default one in the other arithmetic),
        measured the same way (clock warm-up, K steps between synchronisations, median of a few repetitions, event-timed
        kernel pass, oracle check of the timed launch sequence): a side object of the N = 1 line.  `value` and
        `ms_per_step` of the line stay those of the default workload.  Started only while the command's time budget lasts."""
        if left() < need_s:
            return {"skipped": f"time budget ({budget:g} s; --full runs it)"}
        t_side = time.perf_counter()
        try:
            j = Job(name, exact, options)
            sv = Verifier(j, n_random=32 if args.full else 0) if (j.rx and not args.no_verify) else None
            sreps = j.measure(args.steps, args.warmup, 5 if args.full else 3, sv)
            sdt = statistics.median(sreps)
            ks = min(args.steps, 12)
            skt, skern, sdom, sfk = kernel_pass(j, ks)
            if sv:
                sv.checkpoint(with_streams=True)
            fsec = j.full.frame / j.full.fs
            o = {"workload": j.descr, "arithmetic": ARITH[j.exact], "sub_vfos": int(j.st["n_leaves"]), "ms_per_step": round(sdt / args.steps * 1e3, 4),
                 "ms_per_step_min": round(min(sreps) / args.steps * 1e3, 4), "ms_per_step_max": round(max(sreps) / args.steps * 1e3, 4),
                 "value": round(args.steps * j.st["vfo_samples_per_frame"] / sdt / 1e6, 2), "unit": "MSamples/s",
                 "realtime_factor": round(fsec / (sdt / args.steps), 1),
                 "vfos_at_realtime": int(j.st["n_leaves"] * fsec / (sdt / args.steps)),
                 "algorithmic_GBps_whole_frame": round(args.steps * j.st["algorithmic_bytes_per_frame"] / sdt / 1e9, 1),
                 "frame_frac": round(args.steps * j.st["algorithmic_bytes_per_frame"] / sdt / 1e9 / HBM_PEAK_GBS, 4)}
            if options:
                o["options"] = options
            if sdom:
                o["roofline"] = roofline_object(sdom, skt[sdom], ks, sfk, j.st["algorithmic_bytes_per_frame"], 1,
                                                pmc_for(name, j.exact), j.st["mix_chunks_per_frame"], demanded_valu_per_launch(j.topo, j.exact))
            o["kernels"] = {k: v["avg_ms"] for k, v in skern.items()}
            if sv:
                try:
                    o["verified"] = sv.finish(exact=j.exact == 1)
                except Exception as e:
                    o["verified"] = {"ok": None, "error": f"{type(e).__name__}: {e}"}
            j.close()
            return o
        except Exception as e:  # the bench line must still come out
            return {"error": f"{type(e).__name__}: {e}"}
        finally:
            legs["side:" + key] = round(time.perf_counter() - t_side, 2)

    with leg("main_job_create"):
        job = Job(workload)
    topo, rx, st, full, descr = job.topo, job.rx, job.st, job.full, job.descr
    ver = Verifier(job, n_random=64 if args.full else 32) if (rx and not args.no_verify) else None
    with leg("main_measure"):
        reps = job.measure(args.steps, args.warmup, max(1, n_reps), ver)
    dt = statistics.median(reps)
    vfo_samples, alg_bytes, n_leaves = allsum([st["vfo_samples_per_frame"], st["algorithmic_bytes_per_frame"], st["n_leaves"]])

    kt_steps = min(args.steps, 20)
    with leg("kernel_pass"):
        kt, kernels, dom, frame_kernel_ms = kernel_pass(job, kt_steps)
    verified = None
    if ver:
        with leg("main_verify"):
            ver.checkpoint(with_streams=True)
            try:  # (the oracle runs here, on the host, before the legs below put other frames through this receiver)
                verified = ver.finish(exact=arith == 1)
            except Exception as e:
                verified = {"ok": None, "error": f"{type(e).__name__}: {e}"}
    if use_dist:  # every rank checks its own shard; the line reports the worst
        nbad = allsum([0.0 if (verified is None or verified.get("ok")) else 1.0])[0]
        if verified is not None and nbad:
            verified["ok"] = False
            verified["ranks_with_mismatches"] = int(nbad)

    # the reference's CPU path on this box's host cores, rank 0 at N = 1 (before the side workloads: it is part of the contract,
    # they are not)
    cpu = None
    if world == 1 and not args.no_cpu:
        with leg("cpu_baseline"):
            try:
                cpu = cpu_baseline(workload)
            except Exception as e:  # the bench line must still come out
                cpu = {"error": f"{type(e).__name__}: {e}"}

    # third (N = 1): through the C ABI from HOST buffers -- what a Qt / C++ host sees, PCIe both ways.
    abi = None
    if world == 1 and not args.no_abi and rx:
        t_abi = time.perf_counter()
        n_abi = max(8, min(args.steps, 24))
        pay_mb = sum(topo.vfos[i].n_out * 2 if topo.vfos[i].demod_usb else topo.vfos[i].n_stage_out
                     for i in topo.leaves_in_publish_order()) / 1e6

        def host_loop(kind):
            f32 = job.frames_np[0]
            u8 = (f32 + 127).astype(np.uint8)
            pinned = torch.from_numpy(f32.copy()).pin_memory().numpy()
            best = []
            for _ in range(3):
                barrier()
                t0 = time.perf_counter()
                if kind == "sync_pageable":      # sdrx_process per frame: submit + wait, nothing overlaps
                    for _k in range(n_abi):
                        rx.process(f32)
                elif kind == "sync_pinned":
                    for _k in range(n_abi):
                        rx.process(pinned)
                else:                            # the pipelined interface: submit(f+1); wait() -> f
                    src = {"pipelined_pageable": f32, "pipelined_pinned": pinned, "pipelined_u8": u8}[kind]
                    sub = rx.submit_u8 if kind == "pipelined_u8" else rx.submit
                    sub(src)
                    for _k in range(1, n_abi):
                        sub(src)
                        rx.wait()
                    rx.wait()
                barrier()
                best.append((time.perf_counter() - t0) / n_abi)
            return round(statistics.median(best) * 1e3, 4)

        rx.set_publish(False)  # a ctypes callback costs microseconds per leaf: a C host pays nanoseconds
        abi = {"frames": n_abi, "payload_MB_per_frame": round(pay_mb, 2), "input_MB_per_frame": round(topo.frame * 8 / 1e6, 2),
               "note": "host float/byte frame in, int16 payloads in host memory out, publish callbacks off; median of 3"}
        for kind in ("sync_pageable", "sync_pinned", "pipelined_pageable", "pipelined_pinned", "pipelined_u8"):
            abi[kind + "_ms"] = host_loop(kind)
        # dongle bytes WITH the DC-bias removal of the shipped sdr_25E profile (correct_dc_bias=1, sdrj.cpp:271-286), bit for bit
        # the reference's sequentially rounded recurrence.  Input: the capture-like stream (sdrreceiver_amd/synth.py: noise,
        # carriers, bursts and the ADC OFFSET the correction exists for), 8 frames in turn after 8 frames of settling -- the
        # recurrence runs in verified steps of 8 x 1024 samples where it can (k_dc_chain_spec) and sample by sample where it
        # cannot.  `u8_dc_zero_offset_sync_ms`: the zero-mean LCG frames of the other legs, NO offset at all (the estimate wanders
        # through zero, binade after binade).
        cap = synth.capture_like_u8(8, topo.frame, topo.fs) if topo.fs == 1536000 else None
        cap = [cap[2 * topo.frame * f: 2 * topo.frame * (f + 1)] for f in range(8)] if cap is not None else [(job.frames_np[0] + 128).astype(np.uint8)]
        for b_ in cap:
            rx.process_u8(b_, correct_dc=True)
        st0 = rx.stats()
        barrier()
        t1 = time.perf_counter()
        for b_ in cap:
            rx.process_u8(b_, correct_dc=True)
        barrier()
        abi["u8_dc_sync_ms"] = round((time.perf_counter() - t1) / len(cap) * 1e3, 4)
        rx.submit_u8(cap[0], correct_dc=True)
        barrier_t = time.perf_counter()
        for b_ in cap[1:]:
            rx.submit_u8(b_, correct_dc=True)
            rx.wait()
        rx.wait()
        abi["u8_dc_pipelined_ms"] = round((time.perf_counter() - barrier_t) / len(cap) * 1e3, 4)
        abi["u8_dc_ms_per_frame"] = abi["u8_dc_pipelined_ms"]
        st1 = rx.stats()
        abi["u8_dc_blocks"] = {"walked": int(st1["dc_blocks"] - st0["dc_blocks"]), "redone_sequentially": int(st1["dc_fallback_blocks"] - st0["dc_fallback_blocks"]),
                               "taken_again_on_their_own": int(st1["dc_retried_blocks"] - st0["dc_retried_blocks"]),
                               "input": "capture-like stream (offsets +1.3 / -0.7 LSB), frames 8-23 of the run"}
        # the three kernels of the DC-bias removal alone (k_dc_products + k_dc_chain_spec + k_dc_apply), HIP events around the group
        rx.enable_kernel_timing(True)
        for b_ in cap:
            rx.process_u8(b_, correct_dc=True)
        kt_dc = rx.kernel_times().get("k_ingest")
        rx.enable_kernel_timing(False)
        if kt_dc and kt_dc["launches"]:
            abi["u8_dc_kernels_ms"] = round(kt_dc["ms"] / kt_dc["launches"], 4)
        u8z = (job.frames_np[0] + 127).astype(np.uint8)
        rx.process_u8(u8z, correct_dc=True)
        barrier()
        t1 = time.perf_counter()
        for _k in range(4):
            rx.process_u8(u8z, correct_dc=True)
        barrier()
        abi["u8_dc_zero_offset_sync_ms"] = round((time.perf_counter() - t1) / 4 * 1e3, 4)
        rx.set_publish(True)
        t1 = time.perf_counter()
        for _k in range(4):
            rx.process(job.frames_np[0])
        abi["sync_with_python_callbacks_ms"] = round((time.perf_counter() - t1) / 4 * 1e3, 4)
        # the same two loops from a plain C99 host (host/abi_bench.c, a child process: its own context on
        # the same GPU), callbacks on -- what the reference's Qt host would see
        exe = os.path.join(ROOT, "host", "abi_bench")
        if workload == "config3" and os.path.exists(exe):
            import subprocess
            try:
                r = subprocess.run([exe, "1024", str(n_abi), str(local)], capture_output=True, text=True, timeout=180)
                abi["c_host"] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-300:]}
            except Exception as e:
                abi["c_host"] = {"error": f"{type(e).__name__}: {e}"}
        legs["through_abi"] = round(time.perf_counter() - t_abi, 2)

    side = {}
    if world == 1 and workload == "config3" and not args.no_side:
        # In order of importance (the time budget may cut the list short): BASELINE.json's north-star target, the same workload in
        # the TOLERANCE arithmetic north_star allows ("within 1e-5 relative float tolerance"; option exact = 0; each checked against
        # the oracle at that tolerance), config 4, the north-star sentence read literally (10 240 flat leaves at 1.536 MS/s),
        # SURVEY.md 8d's flat variant of config 3, and BASELINE config 5's whole tree (65 536 sub VFOs) on this ONE GPU: the N = 1
        # origin of the strong-scaling curve the N > 1 lines carry as `config5_strong`.  (key, workload, exact, seconds it needs)
        plan = [("north_star_10k", "10k", None, 4.0)]
        if not args.fast:
            plan += [("fast_config3", "config3", 0, 2.5), ("robust_config3", "config3", 2, 2.5)]
        plan += [("config4_256", "config4", None, 2.5), ("flat_10k", "flat10k", None, 6.0)]
        if not args.fast:
            plan += [("fast_10k", "10k", 0, 4.0), ("fast_config4", "config4", 0, 2.5), ("robust_10k", "10k", 2, 4.0), ("robust_config4", "config4", 2, 2.5)]
        plan += [("flat_1024", "flat", None, 3.0), ("config5_64k_one_gpu", "64k", None, 9.0)]
        for key, name, ex, need in plan:
            side[key] = side_reading(key, name, exact=ex, need_s=need)

    # fourth (N = 1, --full): the Qt drop-in -- `class vfo` of the reference's unmodified vfo.h over the adapter
    # (host/qt/vfo_adapter.cpp), driven like sdrj::demodData drives it, transmitData / ZmqPublisher::publish included
    if abi is not None and workload == "config3" and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdropin_sdrx.so")):
        import subprocess
        qt = {}
        modes = (("sync", {}), ("sync_shared_upload", {"SDRX_SHARE_UPLOAD": "1"}), ("pipelined", {"SDRX_PIPELINE": "1"})) if args.full else \
                (("pipelined", {"SDRX_PIPELINE": "1"}), ("sync", {}))
        with leg("qt_adapter"):
            for label, env in modes:
                if left() < 4.0:
                    qt[label + "_ms"] = None
                    continue
                try:
                    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dropin_run.py"), "time", "sdrx", "1024", "12"],
                                       capture_output=True, text=True, timeout=300, env=dict(os.environ, SDRX_DEVICE=str(local), **env))
                    qt[label + "_ms"] = json.loads(r.stdout.strip().splitlines()[-1])["ms_per_frame"] if r.returncode == 0 else {"error": r.stderr[-300:]}
                except Exception as e:
                    qt[label + "_ms"] = {"error": f"{type(e).__name__}: {e}"}
        qt["note"] = ("per frame: process() on both main VFOs of config 3 through the public interface of vfo.h, 1024 x (payload copy into "
                      "transmit_usb + ZmqPublisher::publish) included; one context per main VFO; null = did not fit the time budget (--full)")
        abi["qt_adapter"] = qt

    weak, other_key = None, None
    if world > 1 and workload in ("config3", "config5") and not args.no_side:
        # side reading at N > 1: the other of the two multi-GPU workloads -- BASELINE config 5 (65 536 sub VFOs in total, strong
        # scaling) next to the weak-scaled default, or the other way round with --workload config5
        other, other_key, other_scaling = (("config5", "config5_strong", "strong") if workload == "config3" else
                                           ("config3", "weak_config3", "weak"))
        job.close()
        job = None
        with leg("side:" + other_key):
            try:
                wj = Job(other)
                wreps = wj.measure(args.steps, args.warmup, 7 if args.full else 5)
                wdt = statistics.median(wreps)
                ws, wa, wl = allsum([wj.st["vfo_samples_per_frame"], wj.st["algorithmic_bytes_per_frame"], wj.st["n_leaves"]])
                weak = {"workload": wj.descr, "scaling": other_scaling, "ms_per_step": round(wdt / args.steps * 1e3, 4),
                        "value": round(args.steps * ws / wdt / 1e6, 2), "unit": "MSamples/s",
                        "realtime_factor": round((wj.full.frame / wj.full.fs) / (wdt / args.steps), 1),
                        "algorithmic_GBps_whole_job": round(args.steps * wa / wdt / 1e9, 1), "sub_vfos_total": int(wl)}
                wj.close()
            except Exception as e:
                weak = {"error": f"{type(e).__name__}: {e}"}

    failed = []
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        frame_seconds = full.frame / full.fs
        value = args.steps * vfo_samples / dt / 1e6
        out = {
            "metric": "IQ MSamples/s ingested, summed over VFO chains (1.536 MS/s -> 48/12 kHz USB chain)",
            "build_id": build_id(),
            "value": round(value, 2), "unit": "MSamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if workload == "config5" else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": descr, "name": full.name, "vfos_total": int(len(full.vfos)), "sub_vfos_per_gpu": int(st["n_leaves"]),
                       "frame_cf32": full.frame, "fs": full.fs, "arithmetic": ARITH[arith],
                       "options": user_options,
                       "parallelism": (f"vfo-shard x{world}, raw frames RCCL broadcast ({batch} per collective)" if use_dist else "single GPU"),
                       "launches": ("separate kernels, leaf tail of frame f beside the levels of frame f+1 (2 HIP streams)" if args.pipeline
                                    else "one kernel launch per tree level + leaf tail" if args.no_fuse
                                    else "k_mix_levels, one level per launch + leaf tail" if args.no_frame_pipeline
                                    else "per step ONE k_mix_levels launch (level l works on frame k-l: software pipeline over the frames "
                                         "queued back to back) + the leaf tail of the frame that left the last level")},
            "repetitions": len(reps), "ms_per_step_min": round(min(reps) / args.steps * 1e3, 4),
            "ms_per_step_max": round(max(reps) / args.steps * 1e3, 4),
            "timed_region_ms_total": round(sum(reps) * 1e3, 1),
            "raw_iq_msps": round(args.steps * full.frame / dt / 1e6, 2),
            "vfos_at_realtime": int(n_leaves * frame_seconds / (dt / args.steps)),
            "realtime_factor": round(frame_seconds / (dt / args.steps), 1),
            "algorithmic_GBps_whole_frame": round(args.steps * alg_bytes / dt / 1e9, 1),
            "frame_frac_of_hbm_roofline": round(args.steps * alg_bytes / world / dt / 1e9 / HBM_PEAK_GBS, 4),
        }
        if verified is not None:
            out["verified"] = verified
        if dom:
            pm = pmc_for(workload, arith)
            out["roofline"] = roofline_object(dom, kt[dom], kt_steps, frame_kernel_ms, alg_bytes, world, pm, st["mix_chunks_per_frame"],
                                              demanded_valu_per_launch(topo, arith) if world == 1 else None)
        out["kernels"] = kernels
        if abi:
            out["through_abi"] = abi
            ch = abi.get("c_host") if isinstance(abi.get("c_host"), dict) else {}
            # what a C host sees (host/abi_bench.c) where it ran, else the same loops through ctypes
            out["ms_per_step_through_abi"] = ch.get("sdrx_submit_wait_ms", abi["pipelined_pageable_ms"])
            out["ms_per_step_with_payload_d2h"] = ch.get("sdrx_process_ms", abi["sync_pageable_ms"])
        if use_dist:
            out["rccl_world"] = diag_world  # the world size the collective library reports ...
            out["peer_ok"] = diag_peer      # ... and per rank: can its device reach rank 0's directly (hipDeviceCanAccessPeer)?
        if weak:
            out[other_key] = weak
        if side:
            out.update(side)
        if world == 1 and args.configs1:
            # BASELINE.json configs[1] (32 sub VFOs), the same way, as a side reading: a latency-bound
            # plumbing case on this hardware (three ~10 us launches per frame)
            try:
                t2 = tp.config2()
                rx2 = Receiver.from_topology(t2, device=local, exact=arith)
                rx2.set_stream(stream.cuda_stream)
                st2 = rx2.stats()
                for _ in range(args.warmup):
                    rx2.process_device(job.src.data_ptr(), t2.frame)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    rx2.process_device(job.src.data_ptr(), t2.frame)
                torch.cuda.synchronize()
                d2 = (time.perf_counter() - t0) / args.steps
                rx2.close()
                out["configs1_32_sub_vfos"] = {"ms_per_step": round(d2 * 1e3, 4), "value": round(st2["vfo_samples_per_frame"] / d2 / 1e6, 2),
                                               "unit": "MSamples/s", "realtime_factor": round(frame_seconds / d2, 1),
                                               "algorithmic_GBps_whole_frame": round(st2["algorithmic_bytes_per_frame"] / d2 / 1e9, 1)}
            except Exception as e:
                out["configs1_32_sub_vfos"] = {"error": f"{type(e).__name__}: {e}"}
        if cpu is not None:
            out["cpu_baseline"] = cpu
        out["legs_s"] = legs
        out["wall_s"] = round(time.perf_counter() - t_start, 1)
        write_full(out)
        print(json.dumps(compact_line(out)), flush=True)
        failed = [k for k, v in [("", out)] + [(k, v) for k, v in out.items() if isinstance(v, dict)]
                  if isinstance(v.get("verified"), dict) and v["verified"].get("ok") is False]
    warmer.close()
    if job:
        job.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0 and failed:
        print(f"bench: the timed launch sequence did NOT reproduce the oracle ({failed})", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()

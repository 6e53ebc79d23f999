#!/usr/bin/env python3
"""tools/pmc_summary.py <prof_dir> <out_json> [workload] [exact]
   tools/pmc_summary.py --calib <calib_dir> <out_json>

Condense the rocprofv3 --pmc passes written by tools/profile.sh into per-kernel, per-launch numbers.

* HBM bytes: FETCH_SIZE / WRITE_SIZE are reported in units of 1024 B; on gfx950 FETCH_SIZE counts 128-byte
  requests as 64 bytes, so it is doubled (MI355X_MICROARCH.md, section HBM).  L2 hit rate from TCC_HIT/MISS.
* VALU busy -- calibrated in round 3 against tools/valu_calib.hip (VALU streams of known length and class run
  under the same counters; `--calib` condenses that run into profiles/valu_calibration.json):
    - SQ_ACTIVE_INST_VALU is just SQ_INSTS_VALU on gfx950 (one quad-cycle per instruction, whatever its class);
    - the SIMD issues TWO plain two-operand fp32 instructions in one quad-cycle when it can;
      SQ_ACTIVE_INST_VALU2 counts those quad-cycles, so busy quad-cycles = SQ_INSTS_VALU - SQ_ACTIVE_INST_VALU2;
    - elapsed cycles = SQ_BUSY_CYCLES / 32 shader engines of the SAME pass (GRBM_GUI_ACTIVE / 8 also counts
      several microseconds of dispatch overhead outside the kernel's timestamps: +25 % on a 34 us kernel, which
      together with a numerator from another pass is what made round 2's formula read 1.04 / 1.42);
      valu_busy_raw = 4 (I - V2) / (N_SIMD * cycles);
    - a saturated probe of the kernel's own instruction mix reads 0.94-0.98 raw (loop overhead, packed operands):
      valu_busy = valu_busy_raw / that reading, i.e. the probe reads 1.00 by construction and every other probe
      class within a few per cent of it (the table is in the calibration file).
  Every figure of one kernel comes from ONE pass: numerator, cycles and the dispatch's own duration
  (End_Timestamp - Start_Timestamp of the counter CSV).
bench.py reads the result (profiles/current_pmc.json, profiles/pmc_<workload>.json) for roofline.traffic / valu."""
import collections
import csv
import glob
import json
import os
import sys

N_SIMD = 1024  # 256 CUs x 4
N_SE = 32      # 8 XCDs x 4 shader engines: SQ_BUSY_CYCLES is summed over them

KERNEL_KEYS = {  # (substrings of the demangled names: k_mix_decimate<EXACT, LEVEL, ROT>)
    "k_mix_decimate<true, 1": "k_mix_decimate(sub)", "k_mix_decimate<false, 1": "k_mix_decimate(sub)",
    "k_mix_decimate<true, 0": "k_mix_decimate(level0)", "k_mix_decimate<false, 0": "k_mix_decimate(level0)",
    "k_usb_demod": "k_usb_demod", "k_late_decimate": "k_late_decimate", "k_compress": "k_compress",
    "k_ingest": "k_ingest", "k_mix_levels": "k_mix_levels",
}
# which probe kernel of tools/valu_calib.hip has the instruction mix of which product kernel
PROBE_FOR = {"k_mix_levels": "mixlike", "k_mix_decimate(sub)": "mixlike", "k_mix_decimate(level0)": "mixlike",
             "k_usb_demod": "demodlike", "k_late_decimate": "pk_mul"}
PROBE_NAMES = ["fma32", "mul32", "add32", "mul32_sgpr", "pk_mul", "pk_add", "pk_fma", "fma64", "cvt64", "dpp", "mixlike", "demodlike",
               "add32_vv"]

# VALU wave-instructions k_mix_decimate issues per 1024-sample chunk of a d = 5 sub VFO, counted in the
# source (sdrreceiver_amd/csrc/kernels.hip; a packed v_pk_*_f32 on a (re, im) pair counts as ONE):
INST_MIX_D5 = {
    "nco_replay": 16 * 7,      # cmul 3 + n*n 1 + (x+y) 1 + (1.95-s) 1 + n*norm 1, per table entry
    "mix": 16 * 3,             # cmul per sample
    "stage0_registers": 8 * 11,  # 3 pair sums, 4 products, 3 sums, 0+s -- per output (hb_dot2)
    "stage1_registers": 4 * 11,
    "dpp_halo_moves": 2 * 16,  # v_mov_b32_dpp wave_shr:1, two per shifted complex value, 8 values per stage
    "stages2to4_lds": (2 + 1 + 0.5) * 11,  # 128 / 64 / 32 outputs per chunk over 64 lanes (the last one every second chunk on 64)
    "addressing_loop_stores": 26,          # the rest of the measured ~390: address arithmetic, selects, loop
}


# the same in the tolerance arithmetic (option exact = 0): a table entry and the mixer two packed instructions each (multiply + FMA
# with source modifiers), a half-band output 3 pair sums + 1 product + 3 FMAs
INST_MIX_D5_TOLERANCE = {
    "nco_rotations": 16 * 2,
    "mix": 16 * 2,
    "stage0_registers": 8 * 7,
    "stage1_registers": 4 * 7,
    "dpp_halo_moves": 2 * 16,
    "stages2to4_lds": (2 + 1 + 0.5) * 7,
    "addressing_loop_stores": 26,
}


# the robust arithmetic (option exact = 2): the table entries from the exact recurrence (7 packed instructions per step), the mixer
# and the filters as in the tolerance arithmetic
INST_MIX_D5_ROBUST = dict(INST_MIX_D5_TOLERANCE, nco_rotations=16 * 7)


def key_of(name):
    for k, v in KERNEL_KEYS.items():
        if k in name:
            return v
    return None


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


def read_pass(path, keyfn):
    """One counter CSV = one pass: per kernel the median of every counter over its dispatches and the median
    dispatch duration of THAT pass."""
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        k = keyfn(r["Kernel_Name"])
        if k is None:
            continue
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return {k: {"counters": {n: med(v) for n, v in c.items()}, "dur_us": med(list(dur[k].values())), "dispatches": len(dur[k])}
            for k, c in vals.items()}


def valu_derived(c, dur_us):
    """The VALU / LDS / wait readings of one pass (see the module docstring)."""
    if not all(n in c for n in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU2", "SQ_BUSY_CYCLES")):
        return None
    cycles = c["SQ_BUSY_CYCLES"] / N_SE
    inst, pairs = c["SQ_INSTS_VALU"], c["SQ_ACTIVE_INST_VALU2"]
    d = {"pass_dur_us": round(dur_us, 2), "cycles": int(cycles), "clock_GHz": round(cycles / dur_us / 1e3, 3),
         "valu_insts": int(inst), "dual_issued_frac": round(2 * pairs / max(1.0, inst), 3),
         "valu_busy_raw": round(4 * (inst - pairs) / (N_SIMD * cycles), 3),
         "round2_formula_for_comparison": round(4 * inst / (N_SIMD * cycles), 3)}
    if "SQ_WAVE_CYCLES" in c:
        wc = c["SQ_WAVE_CYCLES"]
        d["wave_cycles_split"] = {n: round(c[n] / wc, 3) for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if n in c}
    if "SQ_ACTIVE_INST_LDS" in c:
        d["lds_inst_busy"] = round(4 * c["SQ_ACTIVE_INST_LDS"] / (N_SIMD * cycles), 3)
    return d


def calib(d, out):
    """valu_calib.hip under the counters: per probe class the raw busy reading of a VALU that is 100 % busy."""
    def probe_key(name):
        if "k_calib<" not in name:
            return None
        return PROBE_NAMES[int(name.split("k_calib<")[1].split(">")[0])]
    res = {"source": os.path.relpath(d), "formula": "valu_busy_raw = 4 (SQ_INSTS_VALU - SQ_ACTIVE_INST_VALU2) / (1024 SIMDs * SQ_BUSY_CYCLES / 32)",
           "probes": {}, "probe_for": PROBE_FOR}
    for f in sorted(glob.glob(os.path.join(d, "probe*_counter_collection.csv"))):
        for k, e in read_pass(f, probe_key).items():
            v = valu_derived(e["counters"], e["dur_us"])
            if v:
                res["probes"][k] = {kk: v[kk] for kk in ("pass_dur_us", "clock_GHz", "valu_insts", "dual_issued_frac", "valu_busy_raw",
                                                         "round2_formula_for_comparison") if kk in v}
                res["probes"][k]["cycles_per_inst"] = round(4.0 / max(1e-9, v["round2_formula_for_comparison"]), 3)
    try:
        res["git_sha"] = open(os.path.join(d, "build_sha.txt")).read().strip()
    except OSError:
        pass
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v["valu_busy_raw"] for k, v in res["probes"].items()}))


def main():
    if sys.argv[1] == "--calib":
        return calib(sys.argv[2], sys.argv[3])
    d, out = sys.argv[1], sys.argv[2]
    workload = sys.argv[3] if len(sys.argv) > 3 else "config3"
    exact = int(sys.argv[4]) if len(sys.argv) > 4 else 1  # option "exact": 1 | 0 tolerance | 2 robust
    passes = [read_pass(f, key_of) for f in sorted(glob.glob(os.path.join(d, "pmc*_counter_collection.csv")))]
    sha = "unknown"
    try:
        sha = open(os.path.join(d, "build_sha.txt")).read().strip()
    except OSError:
        pass
    build_id = None
    for name in ("bench.json", "bench_unprofiled.json"):  # the bench line of the same profile run names the library
        try:
            build_id = json.loads(open(os.path.join(d, name)).read().strip().splitlines()[-1]).get("build_id") or build_id
        except (OSError, ValueError, IndexError):
            pass
    sat = {}
    try:
        cal = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "valu_calibration.json")))
        sat = {k: v["valu_busy_raw"] for k, v in cal["probes"].items()}
    except (OSError, ValueError, KeyError):
        pass
    # average kernel durations of the kernel-trace pass of the same profile directory
    avg_us = {}
    for f in glob.glob(os.path.join(d, "*kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            k = key_of(r["Name"])
            if k:
                avg_us[k] = float(r["AverageNs"]) / 1e3
    res = {"source": os.path.relpath(d), "git_sha": sha, "build_id": build_id, "workload": workload, "exact": exact, "kernels": {},
           "inst_mix": {"k_mix_decimate, d=5 sub VFO, per 1024-sample chunk (source count, packed = 1)": {1: INST_MIX_D5, 0: INST_MIX_D5_TOLERANCE, 2: INST_MIX_D5_ROBUST}[exact],
                        "sum": sum({1: INST_MIX_D5, 0: INST_MIX_D5_TOLERANCE, 2: INST_MIX_D5_ROBUST}[exact].values()),
                        "isa_check": "tools/inst_mix.py: the static v_pk_mul/add/fma_f32 and DPP counts of kernels.s == the source count of the bodies compiled into the kernel (d = 5 leaf: 146/87.5/32/32 executed per chunk)"},
           "note": "per-launch medians; hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE correction); every derived "
                   "figure takes numerator, cycles and duration from ONE pass (tools/pmc_summary.py)"}
    kernels = sorted({k for p in passes for k in p})
    for k in kernels:
        m = {}
        for p in passes:  # merged view (first pass that has a counter wins)
            for n, v in p.get(k, {}).get("counters", {}).items():
                m.setdefault(n, v)
        e = {"counters": {n: round(v, 1) for n, v in m.items()}}
        if k in avg_us:
            e["avg_us"] = round(avg_us[k], 2)
        e["pass_dur_us"] = [round(p[k]["dur_us"], 2) for p in passes if k in p]
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            e["hbm_read_bytes_per_launch"] = int(2 * m["FETCH_SIZE"] * 1024)
            e["hbm_write_bytes_per_launch"] = int(m["WRITE_SIZE"] * 1024)
            e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
        if "TCC_HIT_sum" in m:
            e["l2_hit_rate"] = round(m["TCC_HIT_sum"] / max(1.0, m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 3)
        for p in passes:
            if k in p:
                v = valu_derived(p[k]["counters"], p[k]["dur_us"])
                if v:
                    probe = PROBE_FOR.get(k)
                    if probe in sat:
                        v["calibration_probe"] = probe
                        v["probe_reads_raw"] = sat[probe]
                        v["valu_busy"] = round(v["valu_busy_raw"] / sat[probe], 3)
                    e["valu"] = v
                    break
        res["kernels"][k] = e
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: {"hbm_bytes_per_launch": v.get("hbm_bytes_per_launch"), "l2_hit_rate": v.get("l2_hit_rate"),
                          "valu_busy": v.get("valu", {}).get("valu_busy", v.get("valu", {}).get("valu_busy_raw"))}
                      for k, v in res["kernels"].items()}))


if __name__ == "__main__":
    main()

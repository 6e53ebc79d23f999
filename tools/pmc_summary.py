#!/usr/bin/env python3
"""tools/pmc_summary.py <prof_dir> <out_json> [workload] [exact]

Condense the rocprofv3 --pmc passes written by tools/profile.sh into per-kernel, per-launch numbers:
HBM bytes (FETCH_SIZE / WRITE_SIZE are reported in KiB-ish units of 1024 B; on gfx950 FETCH_SIZE
counts 128-byte requests as 64 bytes, so it is doubled -- MI355X_MICROARCH.md, section HBM), VALU /
LDS / wait counters, L2 hit rate.  bench.py reads the result (profiles/current_pmc.json) for
roofline.traffic."""
import collections
import csv
import glob
import json
import os
import sys

KERNEL_KEYS = {
    "k_mix_decimate<true, 1>": "k_mix_decimate(sub)", "k_mix_decimate<false, 1>": "k_mix_decimate(sub)",
    "k_mix_decimate<true, 0>": "k_mix_decimate(level0)", "k_mix_decimate<false, 0>": "k_mix_decimate(level0)",
    "k_usb_demod": "k_usb_demod", "k_late_decimate": "k_late_decimate", "k_compress": "k_compress",
    "k_ingest": "k_ingest", "k_mix_levels": "k_mix_levels",
}

# VALU wave-instructions k_mix_decimate issues per 1024-sample chunk of a d = 5 sub VFO, counted in the
# source (sdrreceiver_amd/csrc/kernels.hip; a packed v_pk_*_f32 on a (re, im) pair counts as ONE):
INST_MIX_D5 = {
    "nco_replay": 16 * 7,      # cmul 3 + n*n 1 + (x+y) 1 + (1.95-s) 1 + n*norm 1, per table entry
    "mix": 16 * 3,             # cmul per sample
    "stage0_registers": 8 * 11,  # 3 pair sums, 4 products, 3 sums, 0+s -- per output (hb_dot2)
    "stage1_registers": 4 * 11,
    "dpp_halo_moves": 2 * 16,  # v_mov_b32_dpp wave_shr:1, two per shifted complex value, 8 values per stage
    "stages2to4_lds": (2 + 1 + 1) * 11,  # 128 / 64 / 32 outputs per chunk over 64 lanes
    "addressing_loop_stores": 46,        # the rest of the measured ~414: address arithmetic, selects, loop
}


def key_of(name):
    for k, v in KERNEL_KEYS.items():
        if k in name:
            return v
    return None


def main():
    d, out = sys.argv[1], sys.argv[2]
    workload = sys.argv[3] if len(sys.argv) > 3 else "config3"
    exact = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, "pmc*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            k = key_of(r["Kernel_Name"])
            if k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    sha = "unknown"
    try:
        sha = open(os.path.join(d, "build_sha.txt")).read().strip()
    except OSError:
        pass
    # average kernel durations of the kernel-trace pass of the same profile directory
    avg_us = {}
    for f in glob.glob(os.path.join(d, "*kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            k = key_of(r["Name"])
            if k:
                avg_us[k] = float(r["AverageNs"]) / 1e3
    res = {"source": os.path.relpath(d), "git_sha": sha, "workload": workload, "exact": exact, "kernels": {},
           "inst_mix": {"k_mix_decimate, d=5 sub VFO, per 1024-sample chunk (source count, packed = 1)": INST_MIX_D5,
                        "sum": sum(INST_MIX_D5.values()),
                        "isa_check": "tools/inst_mix.py: static v_pk_mul/add/fma_f32 = 148/91/32 and 32 DPP moves in kernels.s == the source count"},
           "note": "per-launch medians; hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE correction)"}
    for k, c in agg.items():
        # median over the dispatches: a few launches of a run differ in shape (pipeline fill / drain)
        m = {n: sorted(v)[len(v) // 2] for n, v in c.items()}
        e = {"counters": {n: round(v, 1) for n, v in m.items()}}
        if k in avg_us:
            e["avg_us"] = round(avg_us[k], 2)
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            e["hbm_read_bytes_per_launch"] = int(2 * m["FETCH_SIZE"] * 1024)
            e["hbm_write_bytes_per_launch"] = int(m["WRITE_SIZE"] * 1024)
            e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
        if "TCC_HIT_sum" in m:
            e["l2_hit_rate"] = round(m["TCC_HIT_sum"] / max(1.0, m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 3)
        res["kernels"][k] = e
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: {x: v.get(x) for x in ("hbm_bytes_per_launch", "l2_hit_rate")} for k, v in res["kernels"].items()}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""tools/pmc_ab_summary.py <pmc_ab_dir> <out.json> -- condenses the counter passes of tools/archive/r3d.sh (default build next to the
experiment builds -DSDRX_GLDS=1|2 and -DSDRX_NT=1, config 3 and 10 240 subs) into one table: per build, workload and
kernel the pass duration, the calibrated VALU busy fraction, the wave-cycle split (SQ_WAIT_ANY / SQ_WAIT_INST_ANY /
SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES), LDS and VMEM instruction counts, HBM-side bytes and the L2 hit rate."""
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_summary as P  # noqa: E402


def main():
    d, out = sys.argv[1], sys.argv[2]
    sat = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "valu_calibration.json")))["probes"]
    res = {}
    for sub in sorted(glob.glob(os.path.join(d, "*_*"))):
        tag = os.path.basename(sub)
        passes = [P.read_pass(f, P.key_of) for f in sorted(glob.glob(os.path.join(sub, "pmc*_counter_collection.csv")))]
        for k in ("k_mix_levels", "k_usb_demod"):
            m = {}
            for p in passes:
                for n, v in p.get(k, {}).get("counters", {}).items():
                    m.setdefault(n, v)
            v = next((x for x in (P.valu_derived(p[k]["counters"], p[k]["dur_us"]) for p in passes if k in p) if x), None)
            if not v:
                continue
            e = {"pass_dur_us": v["pass_dur_us"], "valu_busy": round(v["valu_busy_raw"] / sat[P.PROBE_FOR[k]]["valu_busy_raw"], 3),
                 "valu_insts": v["valu_insts"], "wave_cycles_split": v.get("wave_cycles_split"), "lds_inst_busy": v.get("lds_inst_busy")}
            if "FETCH_SIZE" in m:
                e["hbm_side_read_MB"] = round(2 * m["FETCH_SIZE"] * 1024 / 1e6, 1)
            if "WRITE_SIZE" in m:
                e["hbm_side_write_MB"] = round(m["WRITE_SIZE"] * 1024 / 1e6, 1)
            if "TCC_HIT_sum" in m:
                e["l2_hit_rate"] = round(m["TCC_HIT_sum"] / max(1.0, m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 3)
            for n in ("SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAIT_INST_LDS"):
                if n in m:
                    e[n] = int(m[n])
            res.setdefault(tag, {})[k] = e
    json.dump(res, open(out, "w"), indent=1)
    for tag, ks in res.items():
        for k, e in ks.items():
            print(tag, k, e["pass_dur_us"], e["valu_busy"], e["wave_cycles_split"])


if __name__ == "__main__":
    main()

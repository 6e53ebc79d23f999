#!/bin/bash
# tools/trace_dc_steps.sh -- per-kernel durations (rocprofv3 --kernel-trace --stats) of tools/dc_steps_probe.py: k_dc_products,
# k_dc_chain_spec<NW>, k_dc_apply on its four byte streams.  PER_STEP / SDRX_DC_ROUNDS as for the probe.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_dc_steps; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python3 tools/dc_steps_probe.py 16 > $OUT/probe.txt 2> $OUT/err.txt
grep -v amdgpu.ids $OUT/probe.txt | sed "s/\"k_mix_decimate(level0)\": [0-9.]*, \"k_compress\": [0-9.]*, //"
python3 - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("$OUT/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the probe runs stream after stream, value after value: cut the k_dc_chain_spec launches into groups of 32 (16 settle + 16 timed)
groups = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "k_dc_" in n:
        key = n.split("sdrx::")[1].split("(")[0]
        groups.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in groups.items():
    per = 32
    print(k, len(v), "launches; mean us per run of", per, ":", [round(sum(v[i + 16:i + per]) / 16, 1) for i in range(0, len(v) - per + 1, per)])
PY

#!/bin/bash
# kernel timeline of a few steady-state frames (start/end in us, queue id): do frames overlap?
# usage: tools/trace_overlap.sh [tag] [bench args...]
TAG=${1:-ov}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python3 bench.py --steps 12 --warmup 3 --reps 1 --no-cpu --no-abi $* > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$OUT/*kernel_trace.csv")[0])))
rows = [r for r in rows if "sdrx::k_" in r["Kernel_Name"] and "nco_init" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
first = max(0, n - 36)
t0 = int(rows[first]["Start_Timestamp"])
with open("$OUT/timeline.txt", "w") as f:
    for r in rows[first:first + 24]:
        name = r["Kernel_Name"].split("sdrx::")[1].split("(")[0][:28]
        line = f"{name:30s} q{r['Queue_Id']:>3s} start {(int(r['Start_Timestamp'])-t0)/1e3:8.1f}  end {(int(r['End_Timestamp'])-t0)/1e3:8.1f} us"
        print(line); f.write(line + "\n")
PY

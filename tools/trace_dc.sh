#!/bin/bash
# tools/trace_dc.sh -- kernels AND copies of a few pipelined frames of dongle bytes with the DC-bias removal (do the payload
# copy of frame f and the recurrence of frame f + 1 overlap?): rocprofv3 --kernel-trace --memory-copy-trace of tools/dc_overlap_probe.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_dc; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace -d $OUT -o t --output-format csv -- python3 tools/dc_overlap_probe.py pipelined 12 > $OUT/probe.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob
ev = []
for r in csv.DictReader(open(glob.glob("$OUT/*kernel_trace.csv")[0])):
    if "nco_init" not in r["Kernel_Name"]:  # (every kernel: the runtime's own copy kernels too)
        n = r["Kernel_Name"]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (n.split("sdrx::")[1] if "sdrx::" in n else n).split("(")[0][:26] + " q" + r.get("Queue_Id", "?")))
for f in glob.glob("$OUT/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", ""))[12:34]))
ev.sort()
last = ev[-60:]
t0 = last[0][0]
with open("$OUT/timeline.txt", "w") as f:
    for a, b, n in last:
        line = f"{n:30s} start {(a - t0) / 1e3:9.1f}  end {(b - t0) / 1e3:9.1f}  ({(b - a) / 1e3:7.1f} us)"
        print(line); f.write(line + "\n")
PY

#!/bin/bash
# tools/pmc_quick.sh <tag> "<counters>" [bench args] -- one --pmc pass, per-kernel averages printed
TAG=$1; PMC=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcq_$TAG; mkdir -p $OUT
rocprofv3 --pmc $PMC -d $OUT -o p --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-abi --no-side --no-verify $* > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY

#!/bin/bash
# tools/trace_workloads.sh -- rocprofv3 kernel-trace stats + bench line of the non-default workloads
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for W in flat config2 config4 10k 64k; do
  OUT=gpurun_out/wl_$W; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python3 bench.py --workload $W --steps 12 --warmup 3 --reps 5 --no-cpu --no-abi > $OUT/bench.json 2> $OUT/err.txt
  ls $OUT | tr '\n' ' '; echo
done

#!/bin/bash
# tools/r5_config4_sweep.sh -- config 4 (3 mains + 256 fused /5 leaves): frame time against the number of time segments per
# VFO-frame (option "segments") and the order of the levels in k_mix_levels' list (SDRX_LEVEL_ORDER)
cd "$GRAFT_REPO_ROOT"
for ORD in 0 1; do
  for SEG in 0 6 8 10 12 15 20 30; do
    echo -n "level_order=$ORD segments=$SEG: "
    SDRX_LEVEL_ORDER=$ORD python bench.py --no-cpu --no-abi --no-side --no-verify --steps 40 --reps 5 --workload config4 --segments $SEG 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"
  done
done

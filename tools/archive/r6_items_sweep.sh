#!/bin/bash
# round 6: work items per CU / fewest chunks per segment once more on the final kernels (exact arithmetic)
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --no-cpu --no-abi --no-side --steps 40 --reps 9 $* 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'], {k:v['avg_ms'] for k,v in d['kernels'].items()}, d['verified']['ok'])"; }
for i in 1 2; do
for ipc in 16 24 32 40 48 64; do echo -n "[config3 items/CU $ipc min_seg 4] "; SDRX_ITEMS_PER_CU=$ipc run; done
for ms in 2 3 6; do echo -n "[config3 items/CU 32 min_seg $ms] "; SDRX_MIN_SEG=$ms run; done
echo -n "[config3 items/CU 64 min_seg 2] "; SDRX_ITEMS_PER_CU=64 SDRX_MIN_SEG=2 run
done
for ipc in 32 64 96; do for ms in 4 2; do echo -n "[10k items/CU $ipc min_seg $ms] "; SDRX_ITEMS_PER_CU=$ipc SDRX_MIN_SEG=$ms run --workload 10k; done; done
for ipc in 32 64; do echo -n "[config4 items/CU $ipc] "; SDRX_ITEMS_PER_CU=$ipc run --workload config4; done

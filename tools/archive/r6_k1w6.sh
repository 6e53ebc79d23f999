#!/bin/bash
# round 6: six waves per SIMD for the mix item (-DSDRX_K1_MIN_WAVES=6: 80 VGPRs, 56-100 bytes of scratch) in the three arithmetics
cd "$GRAFT_REPO_ROOT"
A=sdrreceiver_amd/csrc/ab/k1w6.so
for i in 1 2 3; do for ar in 0 2 1; do for v in "" "SDRX_LIB=$A"; do
  echo -n "[config3 arith $ar ${v:-default}] "; env $v python3 bench.py --no-cpu --no-abi --no-side --steps 40 --reps 9 --arith $ar 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'], {k:v['avg_ms'] for k,v in d['kernels'].items()}, d['verified']['ok'])"
done; done; done
for ar in 0 2; do for v in "" "SDRX_LIB=$A"; do
  echo -n "[10k arith $ar ${v:-default}] "; env $v python3 bench.py --no-cpu --no-abi --no-side --steps 40 --reps 9 --arith $ar --workload 10k 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'], {k:v['avg_ms'] for k,v in d['kernels'].items()}, d['verified']['ok'])"
done; done

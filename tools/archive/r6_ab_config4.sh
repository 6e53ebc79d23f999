#!/bin/bash
# round 6: config 4 with the /5 window at an LDS row stride of 17 instead of 18 samples (9.3 KB per wave: 17 waves per CU instead of 16)
cd "$GRAFT_REPO_ROOT"
A=sdrreceiver_amd/csrc/ab/stride17.so
SDRX_LIB=$A python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "config4_256 or fused_late or 54w" 2>&1 | tail -2
for i in 1 2 3; do for v in "" "SDRX_LIB=$A"; do for ar in 1 0; do
  echo -n "[${v:-default} arith $ar] "; env $v python3 bench.py --no-cpu --no-abi --no-side --steps 40 --reps 9 --workload config4 --arith $ar 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'], {k:v['avg_ms'] for k,v in d['kernels'].items()}, d['verified']['ok'])"
done; done; done

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
X=sdrreceiver_amd/csrc/ab/packed_exact.so
SDRX_LIB=$X python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "exact_mode or random_trees_against or capture or longest_audio or out_of_range" 2>&1 | tail -3
for i in 1 2 3; do for w in config3 10k config4; do for v in "" "SDRX_LIB=$X"; do
  echo -n "[$w ${v:-default} exact] "; env $v python3 bench.py --no-cpu --no-abi --no-side --steps 40 --reps 9 --workload $w 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'], {k:v['avg_ms'] for k,v in d['kernels'].items()}, d['verified']['ok'])"
done; done; done

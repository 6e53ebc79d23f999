#!/bin/bash
cd "$GRAFT_REPO_ROOT"
A=sdrreceiver_amd/csrc/ab/before_packed.so; H=sdrreceiver_amd/csrc/ab/packed_hilbert_only.so
for i in 1 2 3; do for w in config3 10k config4; do for v in "" "SDRX_LIB=$H" "SDRX_LIB=$A"; do
  echo -n "[$w ${v:-packed} arith 0] "; env $v python3 bench.py --no-cpu --no-abi --no-side --steps 40 --reps 9 --arith 0 --workload $w 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'], {k:v['avg_ms'] for k,v in d['kernels'].items()}, d['verified']['ok'])"
done; done; done

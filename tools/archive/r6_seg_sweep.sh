#!/bin/bash
# round 6: segments per VFO-frame once more on the final build (exact arithmetic, config 3 and 10 240 subs)
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do for s in 0 4 6 8 12 16; do
  echo -n "[config3 segments $s] "; python3 bench.py --no-cpu --no-abi --no-side --steps 40 --reps 9 --segments $s 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'], {k:v['avg_ms'] for k,v in d['kernels'].items()}, d['verified']['ok'])"
done; done
for s in 0 1 2 4; do
  echo -n "[10k segments $s] "; python3 bench.py --no-cpu --no-abi --no-side --steps 40 --reps 9 --segments $s --workload 10k 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'], {k:v['avg_ms'] for k,v in d['kernels'].items()}, d['verified']['ok'])"
done

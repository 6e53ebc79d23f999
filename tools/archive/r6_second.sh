#!/bin/bash
# round 6: parity of the in-wave demodulation and of the robust arithmetic, then the bench (legs, A/B of fuse_demod)
O=gpurun_out/r6b; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "exact_mode or fast_mode or strong_carrier or random_trees or segmentation or config4_256 or back_to_back or frame_pipeline or decimation_depths" > $O/parity.log 2>&1
echo "parity rc $?"; tail -25 $O/parity.log
python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "strong_carrier" -s 2>&1 | grep -E "worst|xfail|passed|failed" > $O/carrier.log; cat $O/carrier.log
S=$(date +%s.%N); python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; E=$(date +%s.%N)
echo "bench rc $? wall $(echo "$E - $S" | bc) bytes $(wc -c < $O/bench.json)"
cp bench_full.json $O/bench_full.json
python3 -c "import json; d=json.load(open('$O/bench_full.json')); print(d['legs_s'], d['wall_s'], d['ms_per_step'])"
for i in 1 2 3; do
  for o in "" "--option fuse_demod=0"; do
    echo -n "[${o:-default}] "; python3 bench.py --no-cpu --no-abi --no-side --steps 40 --reps 9 $o 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'], {k:v['avg_ms'] for k,v in d['kernels'].items()}, d['verified']['ok'])"
  done
done
for w in 10k config2; do for o in "" "--option fuse_demod=0"; do
    echo -n "[$w ${o:-default}] "; python3 bench.py --no-cpu --no-abi --no-side --steps 40 --reps 9 --workload $w $o 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'], {k:v['avg_ms'] for k,v in d['kernels'].items()}, d['verified']['ok'])"
done; done
for o in "--fast" "--fast --option fuse_demod=0"; do
    echo -n "[$o] "; python3 bench.py --no-cpu --no-abi --no-side --steps 40 --reps 9 $o 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'], {k:v['avg_ms'] for k,v in d['kernels'].items()}, d['verified']['ok'])"
done

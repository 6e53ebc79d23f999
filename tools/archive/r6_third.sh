#!/bin/bash
# round 6: the whole GPU suite with durations, the exhaustive config-5 test once, the driver's bench command (legs), robust rows
O=gpurun_out/r6c; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
S=$(date +%s); python3 -m pytest tests -m gpu -q -x --durations=25 > $O/pytest.log 2>&1; rc=$?; E=$(date +%s)
echo "pytest rc $rc in $((E-S)) s"; tail -45 $O/pytest.log
S=$(date +%s); SDRX_EXHAUSTIVE=1 python3 -m pytest tests/test_gpu_full_size.py -m gpu -q -k all_65536 -s > $O/exhaustive_config5.txt 2>&1; E=$(date +%s)
echo "exhaustive rc $? in $((E-S)) s"; tail -3 $O/exhaustive_config5.txt
S=$(date +%s.%N); python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; rc=$?; E=$(date +%s.%N)
echo "bench rc $rc wall $(python3 -c "print(round($E-$S,1))") bytes $(wc -c < $O/bench.json)"
cp bench_full.json $O/bench_full.json
python3 -c "import json; d=json.load(open('$O/bench_full.json')); print(d['legs_s'], d['wall_s'], d['ms_per_step'])"

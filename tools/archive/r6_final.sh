#!/bin/bash
# round 6, final build: the whole GPU suite (durations), smoke, and the driver's exact bench command
O=gpurun_out/r6z; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
S=$(date +%s); python3 -m pytest tests -m gpu -q -x --durations=12 > $O/pytest.log 2>&1; rc=$?; E=$(date +%s)
echo "pytest rc $rc in $((E-S)) s"; tail -22 $O/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$(date +%s.%N); python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; rc=$?; E=$(date +%s.%N)
echo "bench rc $rc wall $(python3 -c "print(round($E-$S,1))") bytes $(wc -c < $O/bench.json)"
cp bench_full.json $O/bench_full.json
python3 -c "import json; d=json.load(open('$O/bench_full.json')); print(d['legs_s'], d['wall_s'], d['ms_per_step'], d['roofline'].get('pmc_matches_build'))"

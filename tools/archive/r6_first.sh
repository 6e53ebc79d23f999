#!/bin/bash
# round 6, first GPU call: the driver's exact bench command (wall time, line size, legs), then the GPU suite with durations
mkdir -p gpurun_out/r6a
cd "$GRAFT_REPO_ROOT"
python3 -c "import torch" 2>/dev/null
/usr/bin/time -v python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6a/bench.json 2> gpurun_out/r6a/bench.err
echo "bench rc $? bytes $(wc -c < gpurun_out/r6a/bench.json)"
grep -E "Elapsed|Maximum resident" gpurun_out/r6a/bench.err
cp bench_full.json gpurun_out/r6a/ 2>/dev/null
python3 -c "import json; d=json.load(open('bench_full.json')); print(d['legs_s'], d['wall_s'])"
/usr/bin/time -v python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6a/bench2.json 2> gpurun_out/r6a/bench2.err
grep -E "Elapsed" gpurun_out/r6a/bench2.err
python3 -m pytest tests -m gpu -q -x --durations=40 > gpurun_out/r6a/pytest.log 2>&1
echo "pytest rc $?"; tail -60 gpurun_out/r6a/pytest.log

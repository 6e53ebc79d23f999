#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6z; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "rc $? bytes $(wc -c < $O/bench.json)"; cp bench_full.json $O/bench_full.json

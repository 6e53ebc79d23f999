#!/bin/bash
# round 6, final build: what does not fit the default suite / the driver's command -- the exhaustive config-5 parity test, bench.py --full,
# the memory-scale workloads, a longer soak of the exact DC-bias removal
O=gpurun_out/r6x; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
S=$(date +%s); SDRX_EXHAUSTIVE=1 python3 -m pytest tests/test_gpu_full_size.py -m gpu -q -k all_65536 -s > $O/exhaustive_config5.txt 2>&1; echo "exhaustive rc $? in $(( $(date +%s)-S )) s"; tail -2 $O/exhaustive_config5.txt
S=$(date +%s); python3 bench.py --full > $O/bench_full_mode_line.json 2> $O/bench_full_mode.err; echo "bench --full rc $? in $(( $(date +%s)-S )) s, line $(wc -c < $O/bench_full_mode_line.json) bytes"; cp bench_full.json $O/bench_full_mode.json
python3 tests/dc_soak.py 120 7 > $O/dc_soak.txt 2>&1; echo "dc soak rc $?"; tail -2 $O/dc_soak.txt
for w in 256k 768k; do
  S=$(date +%s); timeout 900 python3 bench.py --workload $w --steps 8 --warmup 2 --reps 3 --no-cpu --no-abi --no-side > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc $? in $(( $(date +%s)-S )) s: $(python3 -c "import json;d=json.loads(open('$O/bench_$w.json').read());print(d['ms_per_step'], d['realtime_factor'], d['verified']['ok'], d['config']['sub_vfos_per_gpu'])" 2>&1 | tail -1)"
done

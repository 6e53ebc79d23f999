#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python3 tools/packed_check.py > gpurun_out/packed_new.txt 2>&1; SDRX_LIB=sdrreceiver_amd/csrc/ab/before_packed.so python3 tools/packed_check.py > gpurun_out/packed_old.txt 2>&1
diff gpurun_out/packed_new.txt gpurun_out/packed_old.txt && echo "BIT-IDENTICAL to the scalar-FMA build"; tail -3 gpurun_out/packed_new.txt
python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fast_mode or strong_carrier or tolerance or shipped or capture or demodulation_inside or longest_audio or longer_than" 2>&1 | tail -4
A=sdrreceiver_amd/csrc/ab/before_packed.so
for i in 1 2 3; do for v in "" "SDRX_LIB=$A"; do for ar in 0 2; do
  echo -n "[${v:-packed} arith $ar] "; env $v python3 bench.py --no-cpu --no-abi --no-side --steps 40 --reps 9 --arith $ar 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'], {k:v['avg_ms'] for k,v in d['kernels'].items()}, d['verified']['ok'])"
done; done; done
for w in 10k config4; do for v in "" "SDRX_LIB=$A"; do
  echo -n "[$w ${v:-packed} arith 0] "; env $v python3 bench.py --no-cpu --no-abi --no-side --steps 40 --reps 9 --arith 0 --workload $w 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'], {k:v['avg_ms'] for k,v in d['kernels'].items()}, d['verified']['ok'])"
done; done

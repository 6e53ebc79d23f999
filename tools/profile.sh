#!/bin/bash
# tools/profile.sh <tag> [bench args...] -- rocprofv3 passes of bench.py on the GPU box.
# Writes gpurun_out/prof_<tag>/: kernel-trace stats, then separate --pmc passes (never combined
# with tracing domains).  Copy what should be judged into profiles/.
set -u
TAG=${1:-run}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cp profiles/build_sha.txt "$OUT/build_sha.txt" 2>/dev/null || echo unknown > "$OUT/build_sha.txt"
# (--no-clock-warmup: the throw-away receiver that spins the clock up would put ITS launches into the kernel statistics)
ARGS="--steps 40 --warmup 5 --no-cpu --no-abi --no-side --no-verify --no-clock-warmup $*"
python3 bench.py $ARGS > "$OUT/bench_unprofiled.json" 2> "$OUT/bench_unprofiled.err"
rocprofv3 --kernel-trace --stats -d "$OUT" -o trace --output-format csv -- python3 bench.py $ARGS > "$OUT/bench.json" 2> "$OUT/bench.err"
# counter passes: one small group per run (SQ has 8 slots, TCC 4: FETCH_SIZE costs 3, WRITE_SIZE 2)
i=0
# pass 3 is the VALU pass: instructions, dual-issue pairs, the cycle counter and the wait split of ONE pass
# (tools/pmc_summary.py derives the calibrated valu_busy from it; tools/calib.sh runs the probe under the same set)
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC -d "$OUT" -o pmc$i --output-format csv -- python3 bench.py --steps 6 --warmup 2 --reps 1 --no-cpu --no-abi --no-side --no-verify --no-clock-warmup $* > "$OUT/pmc$i.json" 2> "$OUT/pmc$i.err" || echo "pmc pass $i failed: $PMC" >> "$OUT/errors.txt"
done
ls "$OUT"

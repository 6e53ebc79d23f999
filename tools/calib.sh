#!/bin/bash
# tools/calib.sh [tag] -- the VALU-busy calibration: tools/valu_calib (VALU streams of known
# length and class, by construction 100 % VALU-busy) and bench.py under the SAME two --pmc sets, each pass
# carrying its own cycle counter (GRBM_GUI_ACTIVE, SQ_BUSY_CYCLES) next to the VALU counters, and its own
# dispatch timestamps (the counter CSV has Start/End per dispatch): numerator, cycles and duration always
# come from one pass.  tools/pmc_summary.py --calib turns gpurun_out/calib_<tag>/ into profiles/valu_calibration.json.
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/calib_$TAG
mkdir -p "$OUT"
cp profiles/build_sha.txt "$OUT/build_sha.txt" 2>/dev/null || echo unknown > "$OUT/build_sha.txt"
# set A is pass 3 of tools/profile.sh (the VALU pass); B and C show what the other candidates read on the probe
SETA="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
SETB="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 GRBM_GUI_ACTIVE"
SETC=""
./tools/valu_calib 5 3000 > "$OUT/probe_plain.jsonl" 2> "$OUT/probe_plain.err"
./tools/valu_calib 8 2000 > "$OUT/probe_plain_w8.jsonl" 2>> "$OUT/probe_plain.err"
i=0
for S in "$SETA" "$SETB"; do
  i=$((i+1))
  rocprofv3 --pmc $S -d "$OUT" -o probe$i --output-format csv -- ./tools/valu_calib 5 3000 > "$OUT/probe$i.jsonl" 2> "$OUT/probe$i.err" || echo "probe pass $i failed" >> "$OUT/errors.txt"
  rocprofv3 --pmc $S -d "$OUT" -o bench$i --output-format csv -- python3 bench.py --steps 6 --warmup 2 --reps 3 --no-cpu --no-abi --no-side --no-verify > "$OUT/bench$i.json" 2> "$OUT/bench$i.err" || echo "bench pass $i failed" >> "$OUT/errors.txt"
  rocprofv3 --pmc $S -d "$OUT" -o tenk$i --output-format csv -- python3 bench.py --workload 10k --steps 6 --warmup 2 --reps 3 --no-cpu --no-abi > "$OUT/tenk$i.json" 2> "$OUT/tenk$i.err" || echo "10k pass $i failed" >> "$OUT/errors.txt"
done
ls "$OUT" | tr '\n' ' '

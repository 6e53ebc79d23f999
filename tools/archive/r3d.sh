#!/bin/bash
# round-3 experiment run: NT A/B, and the counter split of the LDS-DMA variant next to the default build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
AB=$PWD/sdrreceiver_amd/csrc/ab
export ABARGS="--no-abi --no-side --reps 9"
echo "== NT config3"; bash tools/ab.sh "" "SDRX_LIB=$AB/libsdrx_nt.so" 2>&1 | grep -v amdgpu.ids
export ABARGS="--no-abi --no-side --reps 5 --workload 10k"
echo "== NT 10k"; bash tools/ab.sh "" "SDRX_LIB=$AB/libsdrx_nt.so" 2>&1 | grep -v amdgpu.ids
SETV="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
SETM="SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES"
for V in default glds1 glds2 nt; do
  for W in config3 10k; do
    OUT=gpurun_out/pmc_ab/${V}_$W; mkdir -p $OUT
    LIB=""; [ $V != default ] && LIB=$AB/libsdrx_$V.so
    i=0
    for S in "$SETV" "$SETM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
      i=$((i+1))
      [ $V = glds2 ] && [ $i -gt 2 ] && continue
      SDRX_LIB=$LIB rocprofv3 --pmc $S -d $OUT -o pmc$i --output-format csv -- python3 bench.py --workload $W --steps 6 --warmup 2 --reps 3 --no-cpu --no-abi --no-side > $OUT/pmc$i.json 2> $OUT/pmc$i.err || echo "fail $V $W $i" >> gpurun_out/pmc_ab/errors.txt
    done
  done
done
ls gpurun_out/pmc_ab

#!/bin/bash
# tools/ab_glds.sh -- the LDS-DMA experiment: parity of the two variants, then interleaved A/B timing
cd "$GRAFT_REPO_ROOT"
for G in 1 2; do
  echo "== parity glds$G"
  SDRX_LIB=$PWD/sdrreceiver_amd/csrc/ab/libsdrx_glds$G.so python -m pytest tests/test_gpu_parity.py -x -q -k "fixtures or live_oracle or segmentation or depths or random_trees or short_chunk or full_size or three_level or five_level or frame_pipeline or u8_ingest" 2>&1 | tail -4
done
export ABARGS="--no-abi --no-side --reps 9"
echo "== config3"; bash tools/ab.sh "" "SDRX_LIB=$PWD/sdrreceiver_amd/csrc/ab/libsdrx_glds1.so" "SDRX_LIB=$PWD/sdrreceiver_amd/csrc/ab/libsdrx_glds2.so"
export ABARGS="--no-abi --no-side --reps 5 --workload 10k"
echo "== 10k"; bash tools/ab.sh "" "SDRX_LIB=$PWD/sdrreceiver_amd/csrc/ab/libsdrx_glds1.so" "SDRX_LIB=$PWD/sdrreceiver_amd/csrc/ab/libsdrx_glds2.so"
export ABARGS="--no-abi --no-side --reps 5 --workload flat"
echo "== flat"; bash tools/ab.sh "" "SDRX_LIB=$PWD/sdrreceiver_amd/csrc/ab/libsdrx_glds1.so" "SDRX_LIB=$PWD/sdrreceiver_amd/csrc/ab/libsdrx_glds2.so"

#!/bin/bash
# tools/ab_lib.sh <lib name in csrc/ab> <workload>... -- parity subset with that build, then interleaved A/B against the working tree
cd "$GRAFT_REPO_ROOT"
AB=$PWD/sdrreceiver_amd/csrc/ab
LIB=$1; shift
SDRX_LIB=$AB/$LIB python -m pytest tests/test_gpu_parity.py -x -q -k "fixtures or live_oracle or segmentation or depths or random_trees or short_chunk or full_size or u8_ingest" 2>&1 | tail -2
for W in "$@"; do
  export ABARGS="--no-abi --no-side --reps 7 --workload $W"
  echo "== $W (working tree / $LIB)"; bash tools/ab.sh "" "SDRX_LIB=$AB/$LIB" 2>&1 | grep -v amdgpu.ids
done

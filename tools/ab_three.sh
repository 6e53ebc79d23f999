#!/bin/bash
# three-way interleaved A/B on a workload: working tree, -DSDRX_FIXED_STAGES=0, csrc/ab/libsdrx_prev.so
cd "$GRAFT_REPO_ROOT"
AB=$PWD/sdrreceiver_amd/csrc/ab
for W in "$@"; do
  export ABARGS="--no-abi --no-side --reps 7 --workload $W"
  echo "== $W (working tree / nofixed / prev)"; bash tools/ab.sh "" "SDRX_LIB=$AB/libsdrx_nofixed.so" "SDRX_LIB=$AB/libsdrx_prev.so" 2>&1 | grep -v amdgpu.ids
done

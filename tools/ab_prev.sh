#!/bin/bash
# interleaved A/B of the working tree's library against csrc/ab/libsdrx_prev.so (the previous commit's build)
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
AB=$PWD/sdrreceiver_amd/csrc/ab
export ABARGS="--no-abi --no-side --reps 9"
echo "== config3 (working tree, then previous commit)"; bash tools/ab.sh "" "SDRX_LIB=$AB/libsdrx_prev.so" 2>&1 | grep -v amdgpu.ids
export ABARGS="--no-abi --no-side --reps 5 --workload 10k"
echo "== 10k"; bash tools/ab.sh "" "SDRX_LIB=$AB/libsdrx_prev.so" 2>&1 | grep -v amdgpu.ids
export ABARGS="--no-abi --no-side --reps 5 --workload config4"
echo "== config4"; bash tools/ab.sh "" "SDRX_LIB=$AB/libsdrx_prev.so" 2>&1 | grep -v amdgpu.ids

#!/bin/bash
# compile-time-depth LDS stages (hb_stage_fixed) vs the generic routine: parity, then interleaved A/B
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py -x -q -k "fixtures or live_oracle or segmentation or depths or random_trees or short_chunk or full_size or three_level or five_level or frame_pipeline" 2>&1 | tail -3
AB=$PWD/sdrreceiver_amd/csrc/ab
export ABARGS="--no-abi --no-side --reps 9"
echo "== config3 (default = fixed stages, then -DSDRX_FIXED_STAGES=0)"; bash tools/ab.sh "" "SDRX_LIB=$AB/libsdrx_nofixed.so" 2>&1 | grep -v amdgpu.ids
export ABARGS="--no-abi --no-side --reps 5 --workload 10k"
echo "== 10k"; bash tools/ab.sh "" "SDRX_LIB=$AB/libsdrx_nofixed.so" 2>&1 | grep -v amdgpu.ids

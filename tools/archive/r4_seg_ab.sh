#!/bin/bash
# tools/r4_seg_ab.sh -- chunks per segment of a fused late decimation (SDRX_LATE_MINSEG), config 4, interleaved
cd "$GRAFT_REPO_ROOT"
export ABARGS="--no-abi --no-side --reps 7 --workload config4"
bash tools/ab.sh "SDRX_LATE_MINSEG=2" "SDRX_LATE_MINSEG=3" "SDRX_LATE_MINSEG=4" "SDRX_LATE_MINSEG=6" "SDRX_LATE_MINSEG=8" 2>&1 | grep -v amdgpu.ids

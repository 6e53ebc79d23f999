#!/bin/bash
# tools/ab_many.sh <workload> <lib>... -- interleaved A/B of the working tree's library against several builds in csrc/ab
cd "$GRAFT_REPO_ROOT"
AB=$PWD/sdrreceiver_amd/csrc/ab
W=$1; shift
ARGS=("")
for L in "$@"; do ARGS+=("SDRX_LIB=$AB/$L"); done
export ABARGS="--no-abi --no-side --reps 7 --workload $W"
echo "== $W (working tree / $*)"; bash tools/ab.sh "${ARGS[@]}" 2>&1 | grep -v amdgpu.ids

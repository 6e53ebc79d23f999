#!/bin/bash
# tools/r5_ablate.sh [bench args] -- what each phase of mix_item costs the launch (builds of tools/ab_build.sh abl_<PHASE>
# "-DSDRX_ABL_<PHASE>": WRONG results by design): k_mix_levels' event-timed average with one phase removed, interleaved with
# the default build, three rounds.
cd "$GRAFT_REPO_ROOT"
AB=$PWD/sdrreceiver_amd/csrc/ab
for round in 1 2 3; do
  for L in "" abl_LOAD abl_CP abl_loadcp abl_CARRY abl_LDS abl_STORE abl_all; do
    if [ -n "$L" ]; then export SDRX_LIB=$AB/$L.so; else unset SDRX_LIB; fi
    echo -n "[${L:-default}] "
    python bench.py --no-cpu --no-abi --no-side --no-verify --steps 40 --reps 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"
  done
done

#!/bin/bash
# tools/r4_nco_ab.sh -- -DSDRX_NCO_SCALAR=1 (plain instead of packed NCO recurrence): parity subset, then interleaved A/B
cd "$GRAFT_REPO_ROOT"
bash tools/ab_lib.sh libsdrx_ncoscalar.so config3 10k flat

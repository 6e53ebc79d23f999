#!/bin/bash
cd "$GRAFT_REPO_ROOT"
L=gpurun_out/r6e_prof2.txt; : > $L
bash tools/profile.sh r06_flat10k --workload flat10k >> $L 2>&1
bash tools/profile.sh r06_64k --workload 64k >> $L 2>&1
tail -3 $L

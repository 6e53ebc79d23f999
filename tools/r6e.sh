#!/bin/bash
# round-6 evidence run: rocprofv3 kernel-trace + PMC passes (tools/profile.sh) of the default workload and the side workloads in the
# exact arithmetic, of config 3 / 10 240 subs / config 4 in the tolerance and in the robust arithmetic, and of config 3 with the
# demodulation inside the mix wave (option fuse_demod: the measured negative); tools/r6_collect.sh then copies what is to be judged
# into profiles/ and regenerates the counter summaries bench.py reads
cd "$GRAFT_REPO_ROOT"
L=gpurun_out/r6e_prof.txt; : > $L
bash tools/profile.sh r06 >> $L 2>&1
bash tools/profile.sh r06_10k --workload 10k >> $L 2>&1
bash tools/profile.sh r06_flat --workload flat >> $L 2>&1
bash tools/profile.sh r06_config4 --workload config4 >> $L 2>&1
bash tools/profile.sh r06_tolerance --fast >> $L 2>&1
bash tools/profile.sh r06_10k_tolerance --workload 10k --fast >> $L 2>&1
bash tools/profile.sh r06_config4_tolerance --workload config4 --fast >> $L 2>&1
bash tools/profile.sh r06_robust --arith 2 >> $L 2>&1
bash tools/profile.sh r06_10k_robust --workload 10k --arith 2 >> $L 2>&1
bash tools/profile.sh r06_config4_robust --workload config4 --arith 2 >> $L 2>&1
bash tools/profile.sh r06_fuse_demod --option fuse_demod=1 >> $L 2>&1
tail -3 $L
# (added later in the round: the two largest side workloads, so that their rows carry a roofline too)
bash tools/profile.sh r06_flat10k --workload flat10k >> $L 2>&1
bash tools/profile.sh r06_64k --workload 64k >> $L 2>&1
tail -3 $L

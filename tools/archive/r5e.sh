#!/bin/bash
# round-5 evidence run: rocprofv3 kernel-trace + PMC passes of the default workload and of the side workloads, exact and
# tolerance arithmetic (tools/profile.sh), then tools/r5_collect.sh copies what is to be judged into profiles/
cd "$GRAFT_REPO_ROOT"
bash tools/profile.sh r05 > gpurun_out/r5e_prof.txt 2>&1
bash tools/profile.sh r05_10k --workload 10k >> gpurun_out/r5e_prof.txt 2>&1
bash tools/profile.sh r05_flat --workload flat >> gpurun_out/r5e_prof.txt 2>&1
bash tools/profile.sh r05_config4 --workload config4 >> gpurun_out/r5e_prof.txt 2>&1
bash tools/profile.sh r05_tolerance --fast >> gpurun_out/r5e_prof.txt 2>&1
bash tools/profile.sh r05_10k_tolerance --workload 10k --fast >> gpurun_out/r5e_prof.txt 2>&1
bash tools/profile.sh r05_config4_tolerance --workload config4 --fast >> gpurun_out/r5e_prof.txt 2>&1
tail -3 gpurun_out/r5e_prof.txt

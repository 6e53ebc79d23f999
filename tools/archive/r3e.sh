#!/bin/bash
# round-3 final evidence run: rocprofv3 kernel-trace + PMC passes of the default workload and of the side workloads
cd "$GRAFT_REPO_ROOT"
bash tools/profile.sh r03 > gpurun_out/r3e_prof.txt 2>&1
bash tools/profile.sh r03_10k --workload 10k >> gpurun_out/r3e_prof.txt 2>&1
bash tools/profile.sh r03_flat --workload flat >> gpurun_out/r3e_prof.txt 2>&1
bash tools/profile.sh r03_config4 --workload config4 >> gpurun_out/r3e_prof.txt 2>&1
bash tools/trace_workloads.sh >> gpurun_out/r3e_prof.txt 2>&1

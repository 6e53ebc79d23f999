#!/bin/bash
# tools/r4_profile_all.sh -- the round's committed profiles: tools/profile.sh (kernel trace + six PMC passes) on the four
# workloads whose counter summaries bench.py reads (profiles/current_pmc.json, pmc_10k / pmc_flat / pmc_config4.json)
cd "$GRAFT_REPO_ROOT"
bash tools/profile.sh r04 > /dev/null 2>&1
bash tools/profile.sh r04_10k --workload 10k > /dev/null 2>&1
bash tools/profile.sh r04_flat --workload flat > /dev/null 2>&1
bash tools/profile.sh r04_config4 --workload config4 > /dev/null 2>&1
for d in r04 r04_10k r04_flat r04_config4; do echo "$d: $(ls gpurun_out/prof_$d | wc -l) files; $(cat gpurun_out/prof_$d/errors.txt 2>/dev/null)"; done
# the exact DC-bias removal's three kernels (tools/dc_time.py: config 1's tree fed dongle bytes with correct_dc)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_r04_dc
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r04_dc -o trace --output-format csv -- python3 tools/dc_time.py > gpurun_out/prof_r04_dc/dc_time.json 2> gpurun_out/prof_r04_dc/err.txt
ls gpurun_out/prof_r04_dc

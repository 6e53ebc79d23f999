#!/bin/bash
# tools/r4_profile_all.sh -- the round's committed profiles: tools/profile.sh (kernel trace + six PMC passes) on the four
# workloads whose counter summaries bench.py reads (profiles/current_pmc.json, pmc_10k / pmc_flat / pmc_config4.json)
cd "$GRAFT_REPO_ROOT"
bash tools/profile.sh r04 > /dev/null 2>&1
bash tools/profile.sh r04_10k --workload 10k > /dev/null 2>&1
bash tools/profile.sh r04_flat --workload flat > /dev/null 2>&1
bash tools/profile.sh r04_config4 --workload config4 > /dev/null 2>&1
for d in r04 r04_10k r04_flat r04_config4; do echo "$d: $(ls gpurun_out/prof_$d | wc -l) files; $(cat gpurun_out/prof_$d/errors.txt 2>/dev/null)"; done

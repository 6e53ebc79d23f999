#!/bin/bash
# tools/r4_collect.sh -- copy what tools/r4_profile_all.sh left in gpurun_out/ into profiles/r04* and regenerate the counter summaries
for d in r04 r04_10k r04_flat r04_config4; do
  mkdir -p profiles/$d
  cp gpurun_out/prof_$d/bench.json gpurun_out/prof_$d/bench_unprofiled.json gpurun_out/prof_$d/build_sha.txt profiles/$d/
  cp gpurun_out/prof_$d/trace_kernel_stats.csv profiles/$d/kernel_stats.csv
  cp gpurun_out/prof_$d/pmc*_counter_collection.csv profiles/$d/
done
mkdir -p profiles/r04_dc
cp gpurun_out/prof_r04_dc/trace_kernel_stats.csv profiles/r04_dc/kernel_stats.csv
cp gpurun_out/prof_r04_dc/dc_time.json profiles/r04_dc/
cp gpurun_out/r4_bench_final.json profiles/r04/bench_default_with_cpu_baseline.json
python tools/pmc_summary.py profiles/r04 profiles/current_pmc.json config3
python tools/pmc_summary.py profiles/r04_10k profiles/pmc_10k.json 10k
python tools/pmc_summary.py profiles/r04_flat profiles/pmc_flat.json flat
python tools/pmc_summary.py profiles/r04_config4 profiles/pmc_config4.json config4

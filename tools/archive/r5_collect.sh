#!/bin/bash
# tools/r5_collect.sh -- copy what tools/r5e.sh left in gpurun_out/ into profiles/r05* and regenerate the counter summaries
set -e
for d in r05 r05_10k r05_flat r05_config4 r05_tolerance r05_10k_tolerance r05_config4_tolerance; do
  [ -d gpurun_out/prof_$d ] || { echo "missing gpurun_out/prof_$d"; continue; }
  mkdir -p profiles/$d
  cp gpurun_out/prof_$d/bench.json gpurun_out/prof_$d/bench_unprofiled.json gpurun_out/prof_$d/build_sha.txt profiles/$d/
  cp gpurun_out/prof_$d/trace_kernel_stats.csv profiles/$d/kernel_stats.csv
  cp gpurun_out/prof_$d/pmc*_counter_collection.csv profiles/$d/
done
python tools/pmc_summary.py profiles/r05 profiles/current_pmc.json config3 1
python tools/pmc_summary.py profiles/r05_10k profiles/pmc_10k.json 10k 1
python tools/pmc_summary.py profiles/r05_flat profiles/pmc_flat.json flat 1
python tools/pmc_summary.py profiles/r05_config4 profiles/pmc_config4.json config4 1
python tools/pmc_summary.py profiles/r05_tolerance profiles/pmc_config3_tolerance.json config3 0
python tools/pmc_summary.py profiles/r05_10k_tolerance profiles/pmc_10k_tolerance.json 10k 0
python tools/pmc_summary.py profiles/r05_config4_tolerance profiles/pmc_config4_tolerance.json config4 0

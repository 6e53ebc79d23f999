#!/bin/bash
# tools/r6_collect.sh -- copy what tools/r6e.sh left in gpurun_out/ into profiles/r06* and regenerate the counter summaries
set -e
for d in r06 r06_10k r06_flat r06_config4 r06_tolerance r06_10k_tolerance r06_config4_tolerance r06_robust r06_10k_robust r06_config4_robust r06_fuse_demod r06_flat10k r06_64k; do
  [ -d gpurun_out/prof_$d ] || { echo "missing gpurun_out/prof_$d"; continue; }
  mkdir -p profiles/$d
  cp gpurun_out/prof_$d/bench.json gpurun_out/prof_$d/bench_unprofiled.json gpurun_out/prof_$d/build_sha.txt profiles/$d/
  cp gpurun_out/prof_$d/trace_kernel_stats.csv profiles/$d/kernel_stats.csv
  cp gpurun_out/prof_$d/pmc*_counter_collection.csv profiles/$d/
done
python tools/pmc_summary.py profiles/r06 profiles/current_pmc.json config3 1
python tools/pmc_summary.py profiles/r06_10k profiles/pmc_10k.json 10k 1
python tools/pmc_summary.py profiles/r06_flat profiles/pmc_flat.json flat 1
python tools/pmc_summary.py profiles/r06_config4 profiles/pmc_config4.json config4 1
python tools/pmc_summary.py profiles/r06_tolerance profiles/pmc_config3_tolerance.json config3 0
python tools/pmc_summary.py profiles/r06_10k_tolerance profiles/pmc_10k_tolerance.json 10k 0
python tools/pmc_summary.py profiles/r06_config4_tolerance profiles/pmc_config4_tolerance.json config4 0
python tools/pmc_summary.py profiles/r06_robust profiles/pmc_config3_robust.json config3 2
python tools/pmc_summary.py profiles/r06_10k_robust profiles/pmc_10k_robust.json 10k 2
python tools/pmc_summary.py profiles/r06_config4_robust profiles/pmc_config4_robust.json config4 2
python tools/pmc_summary.py profiles/r06_fuse_demod profiles/r06_fuse_demod/pmc_summary.json config3 1
[ -d profiles/r06_flat10k ] && python tools/pmc_summary.py profiles/r06_flat10k profiles/pmc_flat10k.json flat10k 1
[ -d profiles/r06_64k ] && python tools/pmc_summary.py profiles/r06_64k profiles/pmc_64k.json 64k 1

#!/bin/bash
# tools/r5_config4_pipeline.sh -- config 4 and config 3: the leaf tail (demodulation) of frame f on a second stream beside the
# levels of frame f + 1 (option "pipeline"), against the one in-order stream; exact and tolerance arithmetic
cd "$GRAFT_REPO_ROOT"
for W in config4 config3; do
  for ARITH in "" "--fast"; do
    for P in "" "--pipeline"; do
      echo -n "$W ${ARITH:-exact} ${P:-one stream}: "
      python bench.py --no-cpu --no-abi --no-side --no-verify --steps 40 --reps 9 --workload $W $ARITH $P 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'], {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
    done
  done
done

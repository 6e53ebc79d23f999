#!/bin/bash
# tools/r4_pair_ab.sh -- SDRX_PAIR_STAGES (stages 3 and 4 of a d = 5 leaf every second chunk): parity subset, then interleaved A/B
# against the build without it (csrc/ab/libsdrx_nopair.so) on config 3 and at 10 240 sub VFOs
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q -m gpu -k "fixtures or live_oracle or segmentation or depths or random_trees or short_chunk or pipeline or 10240 or 1024 or long_run or long_queue" 2>&1 | tail -3
bash tools/ab_many.sh config3 libsdrx_nopair.so
bash tools/ab_many.sh 10k libsdrx_nopair.so

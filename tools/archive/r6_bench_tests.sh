#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests/test_distributed_gpu.py tests/test_bench_contract.py -m gpu -q -x 2>&1 | tail -12

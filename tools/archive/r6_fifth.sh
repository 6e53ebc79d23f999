#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests/test_dropin_qt.py -m gpu -q -x -k "survives or pipelined_two or behind_the_unmodified" 2>&1 | tail -15
python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "submit_wait or payload_copies or random_api" 2>&1 | tail -3

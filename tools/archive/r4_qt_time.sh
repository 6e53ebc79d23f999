#!/bin/bash
# tools/r4_qt_time.sh -- the drop-in tests, then ms per frame through vfo.h on config 3 (three modes, three rounds)
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_dropin_qt.py -x -q 2>&1 | tail -3
for i in 1 2 3; do
  python tools/dropin_run.py time sdrx 1024 12
  SDRX_PIPELINE=1 python tools/dropin_run.py time sdrx 1024 12
done 2>&1 | grep -v amdgpu

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run32
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2

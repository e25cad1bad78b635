#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run17
mkdir -p $O
timeout 1500 python -m pytest tests/test_sharded.py -x -q > $O/pytest.log 2>&1
tail -30 $O/pytest.log

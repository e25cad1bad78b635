#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run21
mkdir -p $O
timeout 1500 python -m pytest tests/test_host_cli.py -x -q -k "multi_process" > $O/pytest_cli.log 2>&1
tail -15 $O/pytest_cli.log
timeout 1500 true
tail -5 $O/pytest_full.log
true
grep "stage ms" $O/bench_200M.err | cut -c1-330; tail -1 $O/bench_200M.log | cut -c1-200

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run22
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_sharded.py -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 900 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --extras 0 > $O/bench_200M.log 2>$O/bench_200M.err
grep "stage ms" $O/bench_200M.err | cut -c1-330; tail -1 $O/bench_200M.log | cut -c1-200

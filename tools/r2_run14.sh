#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run14
mkdir -p $O
timeout 900 python bench.py --steps 2 --warmup 1 --cpu-sample 0 > $O/bench_200M.log 2>&1
grep "stage ms" $O/bench_200M.log | cut -c1-330; tail -1 $O/bench_200M.log | cut -c1-200
timeout 900 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --track-first > $O/bench_200M_track.log 2>&1
grep "stage ms" $O/bench_200M_track.log | cut -c1-330; tail -1 $O/bench_200M_track.log | cut -c1-200
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1
tail -2 $O/pytest.log

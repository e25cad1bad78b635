#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run26
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2
timeout 900 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --extras 0 > $O/bench_200M.log 2>$O/bench_200M.err
grep "stage ms" $O/bench_200M.err | cut -c1-200; tail -1 $O/bench_200M.log | cut -c1-160
timeout 900 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --extras 0 --track-first > $O/bench_200M_track.log 2>$O/bench_200M_track.err
grep "stage ms" $O/bench_200M_track.err | cut -c1-200; tail -1 $O/bench_200M_track.log | cut -c1-160
timeout 900 python bench.py --reads 1000000 --read-len 100 --K 23 --T 2000 --steps 3 --warmup 1 --cpu-sample 0 --extras 0 --pipeline superkmer > $O/bench_C1.log 2>$O/bench_C1.err
tail -1 $O/bench_C1.log | cut -c1-160

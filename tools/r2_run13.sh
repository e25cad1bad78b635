#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run13
mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 900 python bench.py --steps 2 --warmup 1 --cpu-sample 0 > $O/bench_200M.log 2>&1
grep "stage ms" $O/bench_200M.log | cut -c1-330; tail -1 $O/bench_200M.log | cut -c1-200
timeout 900 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --track-first > $O/bench_200M_track.log 2>&1
grep "stage ms" $O/bench_200M_track.log | cut -c1-330; tail -1 $O/bench_200M_track.log | cut -c1-200
timeout 900 python bench.py --reads 50000000 --read-len 250 --K 63 --steps 2 --warmup 1 --cpu-sample 0 > $O/bench_C4.log 2>&1
grep "stage ms" $O/bench_C4.log | cut -c1-330; tail -1 $O/bench_C4.log | cut -c1-200
timeout 900 python bench.py --reads 1000000 --read-len 100 --K 23 --T 2000 --steps 3 --warmup 1 --cpu-sample 0 > $O/bench_C1.log 2>&1
grep "stage ms" $O/bench_C1.log | cut -c1-330; tail -1 $O/bench_C1.log | cut -c1-200
timeout 900 python bench.py --reads 1000000 --read-len 100 --K 23 --T 2000 --steps 3 --warmup 1 --cpu-sample 0 --pipeline direct > $O/bench_C1_direct.log 2>&1
tail -1 $O/bench_C1_direct.log | cut -c1-200

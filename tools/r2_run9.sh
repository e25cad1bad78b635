#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run9
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "golden_case or node_table or saturation or edge_cases or device_resident or first_occurrence or poly_g" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 600 python bench.py --reads 50000000 --steps 2 --warmup 1 --cpu-sample 0 --pipeline superkmer > $O/bench_sk_50M.log 2>&1
grep "stage ms" $O/bench_sk_50M.log | cut -c1-400; tail -1 $O/bench_sk_50M.log | cut -c1-200
timeout 900 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --pipeline superkmer > $O/bench_sk_200M.log 2>&1
grep "stage ms\|node table" $O/bench_sk_200M.log | cut -c1-400; tail -1 $O/bench_sk_200M.log | cut -c1-200
timeout 900 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --pipeline superkmer --track-first > $O/bench_sk_200M_track.log 2>&1
grep "stage ms" $O/bench_sk_200M_track.log | cut -c1-400; tail -1 $O/bench_sk_200M_track.log | cut -c1-200
timeout 900 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --pipeline direct --track-first > $O/bench_direct_200M_track.log 2>&1
tail -1 $O/bench_direct_200M_track.log | cut -c1-200
timeout 900 python bench.py --reads 50000000 --read-len 250 --K 63 --steps 2 --warmup 1 --cpu-sample 0 --pipeline superkmer > $O/bench_sk_C4.log 2>&1
grep "stage ms" $O/bench_sk_C4.log | cut -c1-400; tail -1 $O/bench_sk_C4.log | cut -c1-200
timeout 900 python bench.py --reads 50000000 --read-len 250 --K 63 --steps 2 --warmup 1 --cpu-sample 0 --pipeline direct > $O/bench_direct_C4.log 2>&1
tail -1 $O/bench_direct_C4.log | cut -c1-200

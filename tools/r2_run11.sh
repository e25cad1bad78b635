#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run11
mkdir -p $O
cd soapdenovo-trans_amd/csrc && make -B libsdt_gpu.so EXTRA=-DSDT_SK_TICKS > /dev/null 2>&1; cd ../..
timeout 900 python bench.py --reads 20000000 --read-len 250 --K 63 --steps 1 --warmup 1 --cpu-sample 0 > $O/bench_C4_20M_ticks.log 2>&1
grep "stage ms" $O/bench_C4_20M_ticks.log; tail -1 $O/bench_C4_20M_ticks.log | cut -c1-200
timeout 900 python bench.py --reads 50000000 --steps 1 --warmup 1 --cpu-sample 0 > $O/bench_50M_ticks.log 2>&1
grep "stage ms" $O/bench_50M_ticks.log; tail -1 $O/bench_50M_ticks.log | cut -c1-200

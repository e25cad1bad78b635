#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run2
mkdir -p $O
timeout 600 python bench.py --reads 50000000 --steps 2 --warmup 1 --cpu-sample 0 --pipeline superkmer > $O/bench_sk_50M.log 2>&1
grep "stage ms" $O/bench_sk_50M.log; tail -1 $O/bench_sk_50M.log | cut -c1-200
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o sk50 -- python3 $GRAFT_REPO_ROOT/bench.py --reads 50000000 --steps 2 --warmup 1 --cpu-sample 0 --pipeline superkmer > $GRAFT_REPO_ROOT/$O/bench_sk_50M_prof.log 2>&1)
find $O/prof -name "*kernel_stats*" | head -1 | xargs -I{} cp {} $O/kernel_stats_sk_50M.csv
rm -rf $O/prof
head -25 $O/kernel_stats_sk_50M.csv | cut -c1-160

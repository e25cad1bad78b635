#!/bin/bash
# first GPU contact of the super-k-mer pipeline: parity, then stage timings
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run1
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "golden_case or node_table or saturation or edge_cases or device_resident or first_occurrence or poly_g" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py --reads 20000000 --steps 2 --warmup 1 --cpu-sample 0 --pipeline superkmer > $O/bench_sk_20M.log 2>&1
tail -2 $O/bench_sk_20M.log
timeout 600 python bench.py --reads 20000000 --steps 2 --warmup 1 --cpu-sample 0 --pipeline direct > $O/bench_direct_20M.log 2>&1
tail -1 $O/bench_direct_20M.log
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_sk_50M -o sk50 -- python3 $GRAFT_REPO_ROOT/bench.py --reads 50000000 --steps 2 --warmup 1 --cpu-sample 0 --pipeline superkmer > $GRAFT_REPO_ROOT/$O/bench_sk_50M_prof.log 2>&1)
tail -1 $O/bench_sk_50M_prof.log
find $O/prof_sk_50M -name "*kernel_stats*" | head -1 | xargs -I{} head -30 {}
find $O/prof_sk_50M -type f ! -name "*stats*" -delete

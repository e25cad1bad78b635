#!/bin/bash
# round 6, fourth GPU call: the new tests that failed or were not seen, the tile-1024 variants of the count stage (parity first, then rate)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job4
mkdir -p $O
timeout 900 python3 -m pytest tests/test_spills.py -q > $O/pytest_spills.txt 2>&1; tail -4 $O/pytest_spills.txt
timeout 900 python3 -m pytest tests/test_host_cli.py -q -k "node-limit or one-chunk" > $O/pytest_cli.txt 2>&1; tail -3 $O/pytest_cli.txt
timeout 1200 python3 -m pytest tests/test_fullsize.py -q -s -k "multi_rank" > $O/pytest_multirank.txt 2>&1; grep -h "skew_max_over_mean at\|passed\|failed" $O/pytest_multirank.txt
for v in t1024s1280 t1024s1024; do
  SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_$v.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "node_table_equals_oracle or golden_case or hot_bucket or growth" > $O/pytest_parity_$v.txt 2>&1; echo "$v parity: $(tail -1 $O/pytest_parity_$v.txt)"
done
. tools/ab_env.sh
{
run "SDT_X=0" "--steps 3 --warmup 1"
for v in t1024s1280 t1024s1024; do
  run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_$v.so" "--steps 3 --warmup 1"
done
} 2>&1 | tee $O/ab.txt

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run15
mkdir -p $O
for v in "1024 1536" "768 1536" "1024 1280" "512 1024" "1280 1792"; do
  set -- $v
  (cd soapdenovo-trans_amd/csrc && make -B libsdt_gpu.so EXTRA="-DSDT_SK_FLUSH_AT=$1 -DSDT_SK_MAXFILL=$2" > /dev/null 2>&1)
  timeout 900 python bench.py --steps 2 --warmup 1 --cpu-sample 0 > $O/bench_200M_$1_$2.log 2>&1
  echo "== FLUSH_AT $1 MAXFILL $2"; grep "stage ms" $O/bench_200M_$1_$2.log | cut -c1-250; tail -1 $O/bench_200M_$1_$2.log | cut -c1-120
done

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run25
mkdir -p $O
(cd soapdenovo-trans_amd/csrc && make -B libsdt_gpu.so EXTRA=-DSDT_SK_TICKS > /dev/null 2>&1)
timeout 900 python bench.py --reads 50000000 --steps 1 --warmup 1 --cpu-sample 0 --extras 0 > $O/bench_50M_ticks.log 2>$O/bench_50M_ticks.err
python3 - <<'PY'
import re,sys
t=open("gpurun_out/r2_run25/bench_50M_ticks.err").read()
m=re.search(r"stage ms.*", t); print(m.group(0)[:1200] if m else t[-800:])
PY

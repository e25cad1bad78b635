#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run49
mkdir -p $O
for v in old a_only old a_only; do
cp tools/ab/$v.cuh soapdenovo-trans_amd/csrc/sdt_superkmer_kernels.cuh
(cd soapdenovo-trans_amd/csrc && make -B libsdt_gpu.so > /dev/null 2>&1)
timeout 900 python bench.py --reads 50000000 --steps 3 --warmup 1 --cpu-sample 0 --extras 0 > $O/b.log 2>$O/b.err
echo "$v: $(grep -o 'stage_ms_per_step[^}]*}' $O/b.log) $(grep -o '"value": [0-9.]*' $O/b.log | head -1)"
done

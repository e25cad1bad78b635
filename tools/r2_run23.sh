#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run23
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2
timeout 900 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --extras 0 > $O/bench_200M.log 2>$O/bench_200M.err
grep "stage ms" $O/bench_200M.err | cut -c1-200; tail -1 $O/bench_200M.log | cut -c1-160
for geo in 0 1; do
  (cd soapdenovo-trans_amd/csrc && make -B libsdt_gpu.so EXTRA="-DSDT_SK_TRACK_GEO=$geo" > /dev/null 2>&1)
  timeout 900 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --extras 0 --track-first > $O/bench_200M_track_geo$geo.log 2>$O/bench_200M_track_geo$geo.err
  echo "== track geo $geo"; grep "stage ms" $O/bench_200M_track_geo$geo.err | cut -c1-260; tail -1 $O/bench_200M_track_geo$geo.log | cut -c1-160
done

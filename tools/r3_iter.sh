#!/bin/bash
# one build -> measure iteration on the GPU box: pass-1 parity subset, then the bench line (resident + extras); every step under its own timeout
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_iter; mkdir -p $O
timeout 60 python bench.py --reads 5000000 --cpu-sample 0 --extras 0 --steps 1 --warmup 0 2>&1 | grep "stage ms" | cut -c1-110 || { echo "5M run failed/slow"; exit 1; }
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_fullsize.py -x -q -m gpu -k "golden or node_table or saturation or hot or growth or edge_cases or device_resident or first_occurrence or wide_key or seq_scatter or fullsize or c2 or c4" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 300 python bench.py --cpu-sample 0 ${BENCH_ARGS} > $O/bench.json 2> $O/bench.err
grep "stage ms" $O/bench.err | tail -2
python3 - <<'E'
import json
j=json.loads(open('gpurun_out/r3_iter/bench.json').read().strip().splitlines()[-1])
print({k:j[k] for k in ('value','ms_per_step')}, j['roofline']['stage_ms_per_step'], j['roofline']['frac'], 'track', j.get('track_first') and j['track_first']['value'], 'pcie', j.get('pcie_inclusive') and j['pcie_inclusive']['value'])
E

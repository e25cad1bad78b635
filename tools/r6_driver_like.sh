#!/bin/bash
# what the driver runs at the end of a round, on the tree as committed: the GPU suite with -x, smoke(), the default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_driver_like
mkdir -p $O
[ -n "$SKIP_SUITE" ] || { timeout 1800 python3 -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1; grep -E "passed|failed" $O/pytest_gpu.txt | tail -1; }
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
T0=$(date +%s); timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench.py wall: $(( $(date +%s) - T0 )) s"
python3 - <<'PY'
import json
j=json.loads(open("gpurun_out/r6_driver_like/bench.json").read().strip().splitlines()[-1])
print(round(j["value"]/1e9,2), round(j["ms_per_step"],1), j["roofline"]["frac"], j["roofline"]["traffic"] is not None, j["cpu_baseline"]["kmerfreq_identical"], j["e2e"]["speedup"], j["e2e"]["identical"])
PY

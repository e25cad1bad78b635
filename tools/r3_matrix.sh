#!/bin/bash
# The reporting matrix of BASELINE.json's north star at 1 GPU: K = 23 / 31 / 63 / 95 on 100 / 150 / 250 / 250 bp reads, each as one
# bench.py JSON line (value, roofline, cpu_baseline).  The headline configuration also times the reference on 1/10 of its reads
# with -p <all cores> AND its default -p 8, with the reference's own "time spent on hash reads".  usage: r3_matrix.sh <out dir>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
O=${1:-gpurun_out/r3_matrix}; mkdir -p $O
run() { name=$1; shift; timeout ${MATRIX_TIMEOUT:-900} python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - $O/bench_$name.json $name <<'E'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = j.get("cpu_baseline") or {}
    print(sys.argv[2], round(j["value"] / 1e9, 2), "G k-mers/s", round(j["ms_per_step"], 1), "ms  frac", j["roofline"]["frac"], " track", j["track_first"] and round(j["track_first"]["value"] / 1e9, 2),
          " pcie", j["pcie_inclusive"] and round(j["pcie_inclusive"]["value"] / 1e9, 2), " cpu", c.get("value") and round(c["value"] / 1e6, 1), "M/s p", c.get("cores"), "hash_reads_s", c.get("hash_reads_s"), c.get("p8"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
E
}
run K31_150bp_200M --cpu-sample 20000000 --cpu-p8
run K23_100bp_200M --reads 200000000 --read-len 100 --K 23 --cpu-sample 4000000
run K63_250bp_50M --reads 50000000 --read-len 250 --K 63 --cpu-sample 2000000
run K95_250bp_50M --reads 50000000 --read-len 250 --K 95 --cpu-sample 2000000 --extras 0

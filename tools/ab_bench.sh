#!/bin/bash
# A/B builds of libsdt_gpu.so (gpurun_ab/libsdt_gpu_<name>.so, selected through SDT_GPU_LIB) on one bench configuration.
# usage: ab_bench.sh "<bench.py arguments>" name...      every run under its own timeout
cd "$GRAFT_REPO_ROOT" || exit 1
ARGS=$1; shift
for v in "$@"; do
  printf "%s: " $v
  SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_$v.so timeout ${AB_TIMEOUT:-240} python bench.py --cpu-sample 0 --extras 0 $ARGS 2>&1 | python3 -c '
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        j = json.loads(l); r = j["roofline"]
        print(round(j["value"] / 1e9, 2), "G/s", round(j["ms_per_step"], 1), "ms", r["stage_ms_per_step"], "merges/kmer", r["merges_per_kmer"])
        break
else:
    print("no result (timeout or error)")
'
done

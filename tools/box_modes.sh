#!/bin/bash
# one line per box: tools/box_probe.hip's numbers beside the stage times of the judged workload (bench.py --extras 0 --cpu-sample 0) --
# run as its own gpurun call several times (every call gets a fresh box) and compare the lines (profiles/r5/box_modes.txt)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 -o /tmp/box_probe tools/box_probe.hip 2> /dev/null || exit 1
P1=$(/tmp/box_probe | tail -1)
B=$(timeout 600 python bench.py --extras 0 --cpu-sample 0 2> /dev/null | tail -1 | python3 -c "
import json, sys
j = json.loads(sys.stdin.read()); r = j['roofline']
print(json.dumps({'G_kmers_s': round(j['value'] / 1e9, 2), 'ms_per_step': round(j['ms_per_step'], 1), 'stage_ms': r['stage_ms_per_step']}))")
P2=$(/tmp/box_probe | tail -1)
echo "{\"probe_before\": $P1, \"bench\": $B, \"probe_after\": $P2, \"clocks\": \"$(rocm-smi --showclocks 2> /dev/null | grep -E 'sclk|mclk|fclk' | head -3 | tr -s ' ' | tr '\n' ';')\"}" | tee -a gpurun_out/box_modes.txt

#!/bin/bash
# extracts the replay from graph.c and times variants of its prefetching (run on the GPU box: its host is the target)
set -e
cd "$(dirname "$0")/.."
python3 - <<'PY'
src = open('soapdenovo-trans_amd/csrc/host/graph/graph.c').read()
a = src.index("typedef struct { uint64_t key; uint32_t id, tag; } rent_t;")
b = src.index("static uint64_t home_words(")
open('tools/replay_bench_inc.h', 'w').write(src[a:b])
PY
T=${1:-16}; M=${2:-42000000}
run() { gcc -O2 -pthread "$@" -o /tmp/replay_bench tools/replay_bench.c -lm && echo "== $*" && /tmp/replay_bench $T $M; }
run
ls /sys/devices/system/node | grep -c node; lscpu | grep -i "numa\|model name\|^CPU(s)" ; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null || true
for t in 1 4 8 16; do /tmp/replay_bench $t $M; done
for n in 2 4 8; do echo "interleave over $n nodes"; INTERLEAVE=$n /tmp/replay_bench $T $M; done
rm -f tools/replay_bench_inc.h

#!/bin/bash
# PMC passes of pass 1 on the timed configuration (one rocprofv3 --pmc run per counter, MI355X_MICROARCH.md) + the
# kernel-trace statistics of the same command.  usage: pmc_pipeline.sh <out dir> [bench.py arguments]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
export TMPDIR=/tmp
O=$1; shift
mkdir -p $O
ARGS="--steps 1 --warmup 0 --cpu-sample 0 --extras 0 $*"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/trace -o t -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --extras 0 $* > $ROOT/$O/bench_under_rocprof.json 2> $ROOT/$O/bench_under_rocprof.err)
find $O/trace -name "*kernel_stats*" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/trace
for pmc in FETCH_SIZE WRITE_SIZE TCC_EA0_ATOMIC_sum; do
  for attempt in 1 2 3; do          # (a pass has been seen to stall once: bounded, retried once)
    rm -rf $O/pass_$pmc
    (cd /tmp && timeout 420 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $ROOT/$O/pass_$pmc -o p -- python3 $ROOT/bench.py $ARGS > $ROOT/$O/pass_$pmc.log 2>&1)
    if find $O/pass_$pmc -name "*counter_collection.csv" | grep -q .; then break; fi
  done
done

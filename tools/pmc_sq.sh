#!/bin/bash
# SQ / LDS counter passes of pass 1 on the timed configuration: where do k_sk_count's cycles go?  ("counters
# first".)  One rocprofv3 --pmc run per group of <= 8 SQ counters (MI355X_MICROARCH.md, "rocprofv3 PMC slots").
# usage: pmc_sq.sh <out dir> [bench.py arguments]       summary: tools/pmc_sq_summary.py <out dir> <kmers per pass> <out.json>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
export TMPDIR=/tmp
O=$1; shift
mkdir -p $O
ARGS="--steps 1 --warmup 0 --cpu-sample 0 --extras 0 $*"
G1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
G2="SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC"
G3="SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_ATOMIC_RETURN SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
i=0
for grp in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  for attempt in 1 2; do
    rm -rf $O/pass_$i
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $ROOT/$O/pass_$i -o p -- python3 $ROOT/bench.py $ARGS > $ROOT/$O/pass_$i.log 2>&1)
    if find $O/pass_$i -name "*counter_collection.csv" | grep -q .; then break; fi
  done
done

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run5
mkdir -p $O
timeout 600 python bench.py --reads 50000000 --steps 2 --warmup 1 --cpu-sample 0 --pipeline superkmer > $O/bench_sk_50M.log 2>&1
grep "stage ms" $O/bench_sk_50M.log; tail -1 $O/bench_sk_50M.log | cut -c1-200
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py --reads 50000000 --steps 1 --warmup 0 --cpu-sample 0 --pipeline superkmer > $GRAFT_REPO_ROOT/$O/pmc_$tag.log 2>&1)
  f=$(find $O/pmc_$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:40]
    if "k_sk" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in agg:
    print(k, {c: f"{v:.4g}" for c, v in agg[k].items()})
PY
  else tail -3 $O/pmc_$tag.log; fi
  rm -rf $O/pmc_$tag
done

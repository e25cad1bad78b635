#!/usr/bin/env python3
"""Per-kernel sums of the SQ counter passes of tools/pmc_sq.sh.   usage: pmc_sq_summary.py <dir> <k-mers of one pass> <out.json>
SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md latency table); per-k-mer figures are the raw sums / k-mers."""
import collections, csv, glob, json, sys
d, kmers, out = sys.argv[1], float(sys.argv[2]), sys.argv[3]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for f in glob.glob(f"{d}/pass_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
res = {"kmers_per_pass": kmers, "kernels": {}}
for k, c in sorted(agg.items()):
    if not k.startswith(("k_sk_", "k_count_reads", "k_mark", "k_clear")):
        continue
    e = {"sum": dict(sorted(c.items())), "per_kmer": {n: round(v / kmers, 4) for n, v in sorted(c.items())}}
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        e["share_of_wave_cycles"] = {n: round(c[n] / wc, 4) for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY",
                                                                       "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA") if n in c}
    if c.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_share_of_lds_cycles"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 4)
    res["kernels"][k] = e
json.dump(res, open(out, "w"), indent=1)
for k, e in res["kernels"].items():
    if k.startswith(("k_sk_count", "k_sk_scatter")):
        print(k, json.dumps(e["per_kmer"]), json.dumps(e.get("share_of_wave_cycles")), e.get("lds_conflict_share_of_lds_cycles"))

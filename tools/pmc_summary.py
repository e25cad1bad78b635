#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per pass) for one kernel into a JSON.
usage: pmc_summary.py <pmc_dir> <kernel substring> <kmers processed by those launches> <out.json>"""
import collections
import csv
import glob
import json
import sys

d, kern, kmers, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(f"{d}/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
res = {"kernel": kern, "kmers": kmers, "counters": {k: {"launches": n, "sum": v, "per_launch": v / n, "per_kmer": v / kmers}
                                                   for k, (n, v) in sorted(agg.items())}}
c = res["counters"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  MI355X_MICROARCH.md (HBM): FETCH_SIZE = TCC_EA0_RDREQ x 64 B
    # and under-counts WIDE coalesced streams (128-B requests tallied at 64 B) by 2x; this kernel's reads are
    # random 16-B entry loads (one 64-B request each: RDREQ ~= k-mers), so no doubling is applied.
    fetch = c["FETCH_SIZE"]["sum"] * 1024
    write = c["WRITE_SIZE"]["sum"] * 1024
    res["hbm_bytes"] = {"fetch": fetch, "write": write, "total": fetch + write,
                        "per_kmer": (fetch + write) / kmers,
                        "per_launch": (fetch + write) / c["FETCH_SIZE"]["launches"]}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res.get("hbm_bytes"), indent=1))

#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of one bench.py run (tools/pmc_pipeline.sh) into the JSON bench.py reads for
roofline.traffic.   usage: pmc_pipeline_summary.py <dir with pass_*/> <reads> <read_len> <K> <steps incl. warm-up> <out.json>

FETCH_SIZE / WRITE_SIZE come in KiB.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE tallies the 128-B requests of a wide
coalesced stream (16 B per lane) at 64 B -- such reads are doubled: k_sk_scatter_records and k_sk_count read records that
way (their other reads -- 4-B chunk ids, random 16-B table entries at merge time -- are small next to it, so doubling
over-counts slightly: an upper bound); k_sk_scatter_reads' 4-B-per-lane packed words and the table scans are left as reported."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

d, reads, L, K, steps, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kmers = reads * (L - K + 1) * steps
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"{d}/pass_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
PASS1 = ("k_sk_scatter_reads", "k_sk_scatter_reads_seq", "k_sk_scatter_records", "k_sk_scatter_records_staged", "k_sk_count", "k_sk_count_flat", "k_sk_chunk_place",
         "k_sk_chunk_place_few", "k_sk_scan", "k_sk_seal", "k_sk_init_cursors", "k_count_reads",
         # round 5: the fold of the node log (SDT_FLAG_NODE_LOG)
         "k_bm_finalize", "k_bm_desc_hist", "k_bm_desc_place", "k_bm_class_hist", "k_bm_class_scan", "k_bm_class_place", "k_bm_flat_hist", "k_bm_flat_place")
DOUBLE = ("k_sk_scatter_records", "k_sk_scatter_records_staged", "k_sk_count", "k_sk_count_flat", "k_bm_finalize")
OURS = PASS1 + ("k_clear", "k_mark_hist", "k_delow", "k_export", "k_rehash", "k_fixed_offsets")
res = {"reads": reads, "read_len": L, "K": K, "steps_profiled": steps, "kmers": kmers, "kernels": {}}
tot_f = tot_w = tot_a = 0.0
for k, c in sorted(agg.items()):
    base = k.split("<")[0]
    if base not in OURS:                             # (the workload generator's at::native kernels are not part of the measurement)
        continue
    fetch = c.get("FETCH_SIZE", 0) * 1024 * (2 if base in DOUBLE else 1)
    write = c.get("WRITE_SIZE", 0) * 1024
    atom = c.get("TCC_EA0_ATOMIC_sum", 0)
    res["kernels"][k] = {"fetch_bytes": fetch, "fetch_doubled": base in DOUBLE, "write_bytes": write, "atomics": atom,
                         "bytes_per_kmer": (fetch + write) / kmers, "atomics_per_kmer": atom / kmers}
    if base in PASS1:
        tot_f += fetch; tot_w += write; tot_a += atom
res["complete"] = tot_f > 0 and tot_w > 0          # a pass that never finished (rocprofv3 stalls now and then) must not pass for a measurement
res["hbm_bytes_per_kmer"] = (tot_f + tot_w) / kmers
res["fetch_bytes_per_kmer"] = tot_f / kmers
res["write_bytes_per_kmer"] = tot_w / kmers
res["atomics_per_kmer"] = tot_a / kmers
h = hashlib.sha256()
cs = os.path.join(ROOT, "soapdenovo-trans_amd", "csrc")
for f in sorted(os.listdir(cs)):
    if f.endswith((".cuh", ".hip")):
        h.update(open(os.path.join(cs, f), "rb").read())
res["kernel_source_id"] = h.hexdigest()[:16]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("hbm_bytes_per_kmer", "fetch_bytes_per_kmer", "write_bytes_per_kmer", "atomics_per_kmer", "kernel_source_id")}))

#!/usr/bin/env python3
"""bench.py -- pregraph k-mers hashed per second on MI355X (BASELINE.json metric).

A step = one full pass of the hot path over the workload: reset the node table, chop + insert/count every
read (sdt_gpu_count_reads_device; N>1: extract_route -> RCCL all-to-all -> insert_records), drain,
then the linear-mark + kmerFreq scan.  Inputs (packed 2-bit reads) are resident in HBM before the timed
region.  value = k-mer occurrences of the WHOLE job / wall time of the step (max over ranks).

Workload: BASELINE.json metric "200M x 150bp, K=31" (configs[2]) when --reads is not given, as STRONG
scaling: the same 200 M reads are split over the N ranks.  --reads/--read-len/--K/--T select other configs
(configs[1] = --reads 50000000).

One JSON line on rank 0.  roofline.achieved uses SURVEY.md 8(d)'s algorithmic bytes per k-mer occurrence,
B = 0.25*L/(L-K+1) + 2*E (E = 24/32/48 B reference node), times the k-mers of the chop+insert launches,
divided by their HIP-event time on the library's stream (sdt_gpu_kernel_time).
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def algorithmic_bytes_per_kmer(L, K):
    E = 24 if K <= 31 else (32 if K <= 63 else 48)
    return 0.25 * L / (L - K + 1) + 2 * E


def pmc_traffic_per_kmer(K, kernel="k_count_reads"):
    """HBM bytes per k-mer of the dominant kernel from the PMC passes committed under profiles/ (FETCH_SIZE and
    WRITE_SIZE are collected in separate rocprofv3 --pmc runs, tools/pmc_summary.py; they cannot be read live).
    Only a profile taken with the same key width (file name ..._k<K>.json) counts.
    Returns (bytes per k-mer, source file) or (None, None)."""
    import glob
    import re
    words = lambda k: 1 if k <= 31 else (2 if k <= 63 else 4)
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", f"pmc_{kernel}*.json"))):
        m = re.search(r"_k(\d+)\.json$", f)
        if not m or words(int(m.group(1))) != words(K):
            continue
        try:
            j = json.load(open(f))
            best = (j["hbm_bytes"]["per_kmer"], os.path.relpath(f, ROOT))
        except Exception:
            pass
    return best or (None, None)


def usable_cpus():
    """online CPUs capped by the cgroup v2 quota (the GPU boxes give 16 CPUs of a 256-thread host)"""
    n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except Exception:
        pass
    return n


def cpu_baseline(words_dev, n_reads_total, L, K, sample_reads, log):
    """Rank 0, N=1 only.  Time the reference binary (oracle/_ref, kind 'reference') on the first
    `sample_reads` reads of the same workload, from process start until <prefix>.kmerFreq is complete
    (= parse + chop + hash + mark, the part of pregraph this repo replaces); fall back to the oracle port."""
    import oracle_binding as ob

    n = min(sample_reads, n_reads_total)
    nw = (n * L + 15) // 16
    hw = words_dev[:nw].cpu().numpy().view(np.uint32)

    def unpack(r0, r1):
        """codes of reads [r0, r1) from the packed stream"""
        idx = np.arange(r0 * L, r1 * L, dtype=np.int64)
        return ((hw[idx >> 4] >> (30 - 2 * (idx & 15)).astype(np.uint32)) & 3).astype(np.uint8)

    kmers = n * (L - K + 1)
    cores = usable_cpus()
    exe = ob.ref_binary(31 if K <= 31 else 127)
    if exe and os.access(exe, os.X_OK):
        from soapdenovo_trans_amd import synth
        tmp = tempfile.mkdtemp(prefix="sdt_cpu_")
        try:
            fq = os.path.join(tmp, "reads.fq")
            with open(fq, "wb") as fo:
                qual = b"I" * L
                for r0 in range(0, n, 250_000):
                    r1 = min(n, r0 + 250_000)
                    letters = synth.BASES[unpack(r0, r1)].reshape(r1 - r0, L)
                    fo.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (r0 + i, letters[i].tobytes(), qual) for i in range(r1 - r0)))
            if os.path.getsize(fq) % 32768 == 0:       # reference hangs on exact multiples (survey q9)
                with open(fq, "ab") as fo:
                    fo.write(b"\n")
            synth.write_config(os.path.join(tmp, "lib.cfg"), L, fastq=[fq])
            p_threads = min(cores, 64)
            out = os.path.join(tmp, "out")
            t0 = time.time()
            proc = subprocess.Popen([exe, "pregraph", "-s", os.path.join(tmp, "lib.cfg"), "-K", str(K), "-p",
                                     str(p_threads), "-o", out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            kf = out + ".kmerFreq"
            t1 = None
            while proc.poll() is None and time.time() - t0 < 600:
                if os.path.exists(kf) and os.path.getsize(kf) > 0:
                    with open(kf, "rb") as fi:
                        if fi.read().count(b"\n") >= 255:
                            t1 = time.time()
                            break
                time.sleep(0.02)
            if proc.poll() is None:
                proc.kill()           # exact child we started; the later phases are out of scope here
            proc.wait()
            if t1 is not None:
                return {"value": kmers / (t1 - t0), "unit": "kmers/s", "cores": p_threads, "kind": "reference",
                        "sample": f"first {n} reads ({kmers} k-mers) of the workload as FASTQ; reference "
                                  f"SOAPdenovo-Trans pregraph -K {K} -p {p_threads}, process start until "
                                  f"*.kmerFreq written ({t1 - t0:.2f} s)"}
            log("reference binary did not produce kmerFreq; falling back to the oracle port")
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    n = min(n, 300_000)                      # the single-threaded port: keep it to some tens of seconds
    kmers = n * (L - K + 1)
    codes = unpack(0, n)
    o = ob.Oracle(K, nsets=8)
    offs = (np.arange(n + 1, dtype=np.uint64) * L)
    t0 = time.time()
    o.add_reads(codes, offs)
    o.mark()
    dt = time.time() - t0
    return {"value": kmers / dt, "unit": "kmers/s", "cores": 1, "kind": "port",
            "sample": f"first {n} reads ({kmers} k-mers), oracle/sdt_oracle.c single thread ({dt:.2f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=200_000_000, help="reads of the whole job")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--K", type=int, default=31)
    ap.add_argument("--T", type=int, default=20000, help="synthetic transcripts")
    ap.add_argument("--err", type=float, default=0.002)
    ap.add_argument("--sigma", type=float, default=2.0, help="log-normal sigma of the expression weights (SURVEY C5: 2.5)")
    ap.add_argument("--d", type=int, default=0, help="-d: also run the low-coverage filter (k_delow) in every step")
    ap.add_argument("--cpu-sample", type=int, default=4_000_000, help="reads timed on the CPU baseline (0 = skip)")
    ap.add_argument("--route-batch", type=int, default=2_000_000, help="reads per all-to-all round (N>1)")
    ap.add_argument("--est-distinct", type=int, default=0)
    ap.add_argument("--shard-mode", choices=["filter", "route"], default="filter",
                    help="N>1: 'filter' = every rank holds all reads and inserts only the k-mers it owns (no exchange); "
                         "'route' = reads are split and records travel in an RCCL all-to-all")
    ap.add_argument("--pipeline", choices=["auto", "direct", "superkmer"], default="auto",
                    help="pass-1 kernel family: 'direct' = one device atomic per occurrence (k_count_reads); 'superkmer' = "
                         "minimizer buckets of super-k-mers counted in LDS (k_sk_*); 'auto' = the library's default")
    ap.add_argument("--track-first", action="store_true", help="SDT_FLAG_TRACK_FIRST: what the five-file pipeline runs with")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the N>1 code path (extract_route -> all-to-all -> insert_records) even with one rank")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from soapdenovo_trans_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    dist = None
    sharded_path = world > 1 or args.force_sharded
    if sharded_path:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # SDT_BENCH_SHARE_DEVICE=1 (validation only, never for a reported number): all ranks on cuda:0 with a gloo
        # control plane, so that the N>1 code path can be exercised on a 1-GPU box (RCCL refuses two ranks per device)
        share = os.environ.get("SDT_BENCH_SHARE_DEVICE") == "1"
        if share:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    def log(*a):
        if rank == 0:
            print("[bench]", *a, file=sys.stderr, flush=True)

    K, L = pkg.clamp_K(args.K), args.read_len
    n_total = args.reads
    route = args.shard_mode == "route"
    if sharded_path and not route:
        n_local = n_total                     # owner-filter sharding: the reads are replicated, the TABLE is sharded
    else:
        n_local = n_total // world + (1 if rank < n_total % world else 0)
    kmers_total = n_total * (L - K + 1)
    t0 = time.time()
    words, offsets, nwords = synth.torch_workload(n_local, L, args.T, dev, err=args.err, sigma=args.sigma,
                                                  seed=42 + (1000 * rank if (sharded_path and route) else 0))
    torch.cuda.synchronize()
    log(f"workload: {n_local} reads x {L} bp on rank 0 ({nwords * 4 / 1e9:.2f} GB packed), generated in {time.time() - t0:.1f} s")

    # distinct k-mers ~ true k-mers + errors * K; table sized so that MAX_LOAD is not hit
    # (measured: 0.68 G nodes for 200 M x 150 bp at err 0.002 -- most erroneous k-mers of a highly expressed transcript recur)
    est = args.est_distinct or int(args.T * 2250 + n_total * L * args.err * K * 0.35) // world + (1 << 20)
    flags = {"auto": 0, "direct": pkg.SDT_FLAG_DIRECT, "superkmer": pkg.SDT_FLAG_PARTITION}[args.pipeline]
    if args.track_first:
        flags |= pkg.SDT_FLAG_TRACK_FIRST
    g = pkg.PregraphGPU(K, est_distinct=est, device=dev.index or 0, flags=flags)
    stream = torch.cuda.Stream(device=dev)
    g.set_stream(stream.cuda_stream)
    log(f"node table: {g.table_slots()} slots")

    sharded = None
    if sharded_path:
        from soapdenovo_trans_amd.sharding import ShardedCounter, allreduce_stats
        if route:
            sharded = ShardedCounter(g, world, L, min(args.route_batch, n_local), dev)
        else:
            g.set_owner_filter(rank, world)

    local_inserted = [0]

    def one_step(verify=False):
        g.reset()
        if not sharded_path or not route:
            g.count_reads_device(words, nwords, offsets, n_local, L)
        else:
            with torch.cuda.stream(stream):
                sharded.count_reads(words, nwords, offsets, n_local, verify=verify)
        kmers, nodes = g.finish_count()
        local_inserted[0] = kmers              # this rank's share (owner filter / routed records) before the all-reduce
        if args.d:
            g.delow(args.d)
        hist, linear = g.mark_and_hist()
        if sharded_path:
            hist, kmers, nodes, linear = allreduce_stats(hist, kmers, nodes, linear, dev)
        return kmers, nodes, linear, hist

    def barrier():
        torch.cuda.synchronize()
        if sharded_path:
            dist.barrier()
        torch.cuda.synchronize()

    res = None
    for _ in range(args.warmup):
        res = one_step(verify=True)        # checksum the exchange once, outside the timed region
    g.kernel_time(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = one_step()
    barrier()
    dt = time.perf_counter() - t0
    if sharded_path:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    kmers, nodes, linear, hist = res
    assert kmers == kmers_total, f"processed {kmers} k-mers, expected {kmers_total}"
    assert int(hist.sum()) == nodes, "kmerFreq bins do not add up to the node count"
    stage_ms, sk_counters = g.stage_times()
    kms, launches, _ = g.kernel_time(reset=True)
    log("stage ms per step [direct, sk scatter, sk split, sk count]:", [round(x / args.steps, 2) for x in stage_ms], sk_counters)
    ms_per_step = dt / args.steps * 1e3
    value = kmers_total * args.steps / dt
    B = algorithmic_bytes_per_kmer(L, K)
    # kernel-level: this rank's k-mers over this rank's kernel time (N=1: whole job)
    # N > 1, owner filter: rank 0's chop+insert launches walk ALL reads and insert its share of the k-mers; the bytes
    # that count are those of the k-mers it inserted
    filter_path = sharded_path and not route
    local_kmers = n_local * (L - K + 1) if not sharded_path else (local_inserted[0] if filter_path else None)
    roof = None
    if local_kmers and kms > 0:
        ach = B * local_kmers * args.steps / (kms * 1e-3) / 1e9
        tpk, tsrc = pmc_traffic_per_kmer(K) if not sharded_path else (None, None)
        per_launch_kmers = local_kmers * args.steps / max(launches, 1)
        roof = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5),
                "traffic": None if tpk is None else round(tpk * per_launch_kmers),
                "traffic_unit": "HBM bytes per launch (FETCH_SIZE+WRITE_SIZE PMC passes)", "traffic_source": tsrc,
                "algorithmic_bytes_per_launch": round(B * per_launch_kmers), "kernel": "k_count_reads",
                "bytes_per_kmer": round(B, 3), "launches": int(launches),
                "avg_launch_ms": round(kms / max(launches, 1), 4), "kernel_ms_per_step": round(kms / args.steps, 3)}
        if sharded_path:
            roof["rank"] = 0
            roof["note"] = "rank 0 only: its launches chop every read and insert the k-mers it owns"
    out = {
        "metric": "pregraph k-mers hashed/sec", "value": value, "unit": "kmers/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"{n_total} x {L} bp synthetic transcriptome reads (T={args.T}, err={args.err}), "
                               f"K={K}, pass-1 chop+hash+count" + (f"+delow(-d {args.d})" if args.d else "") + "+kmerFreq"
                               + (f", sigma={args.sigma}" if args.sigma != 2.0 else ""), "reads": n_total, "read_len": L, "K": K,
                   "kmers": kmers_total, "distinct_nodes": nodes, "linear_nodes": linear,
                   "parallelism": (f"owner-sharded table x{world}, " + ("records routed by RCCL all-to-all" if route else
                                   "reads replicated, owner filter (no data-path collective)")) if sharded_path
                   else "single-GPU table"},
        "roofline": roof,
    }
    if rank == 0 and world == 1 and not sharded_path and args.cpu_sample > 0:
        try:
            out["cpu_baseline"] = cpu_baseline(words, n_local, L, K, args.cpu_sample, log)
        except Exception as e:     # the baseline is reported, never required
            log("cpu baseline failed:", repr(e))
            out["cpu_baseline"] = None
    else:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    g.close()
    if sharded_path:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

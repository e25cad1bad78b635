#!/usr/bin/env python3
"""bench.py -- pregraph k-mers hashed per second on MI355X (BASELINE.json metric).

A step = one full pass of the hot path over the workload: reset the node table, chop + insert/count every
read (sdt_gpu_count_reads_device; N>1: sdt_gpu_count_reads_sharded = chop + minimizer buckets -> super-k-mer chunks to
the ranks that own their buckets by grouped ncclSend/ncclRecv over xGMI -> split + count in LDS), drain,
then the linear-mark + kmerFreq scan.  Inputs (packed 2-bit reads) are resident in HBM before the timed
region.  value = k-mer occurrences of the WHOLE job / wall time of the step (max over ranks).

Workload: BASELINE.json metric "200M x 150bp, K=31" (configs[2]) when --reads is not given, as STRONG
scaling: the same 200 M reads -- the very reads of the 1-rank run, rank r takes slice r -- are split over the N ranks
(kmerfreq_sha1 is the same at every N).  --reads/--read-len/--K/--T select other configs
(configs[1] = --reads 50000000).

One JSON line on rank 0.  roofline.achieved uses SURVEY.md 8(d)'s algorithmic bytes per k-mer occurrence,
B = 0.25*L/(L-K+1) + 2*E (E = 24/32/48 B reference node), times the k-mers of pass 1, divided by the HIP-event
time of pass 1's kernels on the library's stream (sdt_gpu_kernel_time; the locality pipeline is three kernels --
scatter, split, count -- every k-mer goes through all of them, so their times add).
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


def node_bytes(K, variant=None):
    """E of SURVEY 8(d): sizeof(kmer_t) of the reference build that holds the key -- 24 / 32 / 48 B for the 31mer / 63mer / 127mer
    variant (inc/newhash.h:65-77).  By default the SMALLEST variant that holds K (= the key words this library computes with);
    `variant` prices another build (SURVEY's C4 runs K = 63 in the 127mer build: E = 48)."""
    v = variant or (31 if K <= 31 else (63 if K <= 63 else 127))
    return {31: 24, 63: 32, 127: 48}[v]


def algorithmic_bytes_per_kmer(L, K, variant=None):
    return 0.25 * L / (L - K + 1) + 2 * node_bytes(K, variant)


def kernel_source_id():
    """hash of the device sources: a PMC profile only counts for the kernels it was taken with"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "soapdenovo-trans_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".cuh", ".hip")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic_per_kmer(K, L, reads):
    """HBM bytes per k-mer of pass 1 from the PMC passes committed under profiles/ (FETCH_SIZE and WRITE_SIZE are
    collected in separate rocprofv3 --pmc runs of THIS bench, tools/pmc_pipeline.sh; they cannot be read live).  Only a
    profile of the same workload taken with the same device sources counts.  Returns (bytes per k-mer, file) or (None, None)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_pass1_*.json")), reverse=True):
        try:
            j = json.load(open(f))
            if ((j["K"], j["read_len"], j["reads"]) == (K, L, reads) and j.get("kernel_source_id") == kernel_source_id()
                    and j.get("complete", True)):
                return j["hbm_bytes_per_kmer"], os.path.relpath(f, ROOT)
        except Exception:
            pass
    return None, None


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


def cpu_baseline(words_dev, n_reads_total, L, K, sample_reads, log, also_p8=False, e2e=True, gpu_hist=None, ours_exe=None):
    """Rank 0, N=1 only.  Returns (cpu_baseline, e2e).

    cpu_baseline: the reference binary (oracle/_ref, kind 'reference') on the first `sample_reads` reads of the same workload,
    timed from process start until <prefix>.kmerFreq is complete (= parse + chop + hash + mark, the part of pregraph the bench's
    step replaces); the GPU's 257 bins for the SAME reads are compared with the file the reference wrote
    (`kmerfreq_identical`).  Falls back to the oracle port when the binary is not there.

    e2e (BASELINE.json north_star: ">= 10x the reference CPU pregraph wall-clock ... with bit-identical *.kmerFreq"; the reference's
    phases: pregraph.c:61-110): the same reference run is left to finish (all five files), `sdt-pregraph` runs on the same
    config, and the five files are compared byte for byte (edge.gz after gunzip: the gzip header carries no data)."""
    import gzip
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
    variant = 31 if K <= 31 else 127
    exe = ob.ref_binary(variant)
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
            cfg = os.path.join(tmp, "lib.cfg")
            synth.write_config(cfg, L, fastq=[fq])

            def run_ref(p_threads, to_the_end):
                """one run of the reference: (seconds from process start until *.kmerFreq is complete, the reference's own
                'time spent on hash reads' in whole seconds or None, seconds until the process ended or None, its stdout,
                the bytes of its *.kmerFreq)"""
                out = os.path.join(tmp, f"ref_p{p_threads}")
                log_path = out + ".stdout"
                t0 = time.time()
                t1 = t2 = None
                with open(log_path, "wb") as lo:
                    proc = subprocess.Popen([exe, "pregraph", "-s", cfg, "-K", str(K), "-p",
                                             str(p_threads), "-o", out], stdout=lo, stderr=subprocess.DEVNULL)
                    kf = out + ".kmerFreq"
                    while proc.poll() is None and time.time() - t0 < 1200:
                        if t1 is None and os.path.exists(kf) and os.path.getsize(kf) > 0:
                            with open(kf, "rb") as fi:
                                if fi.read().count(b"\n") >= 255:
                                    t1 = time.time()
                                    if not to_the_end:
                                        break
                        time.sleep(0.02)
                    if proc.poll() is None:
                        proc.kill()           # exact child we started (the later phases are not wanted, or it hangs)
                    elif proc.returncode == 0:
                        t2 = time.time()
                        if t1 is None:
                            t1 = t2
                    proc.wait()
                hash_s = None
                text = open(log_path, errors="replace").read()
                for line in text.splitlines():
                    if line.startswith("time spent on hash reads:"):       # prlHashReads.c:623 (whole seconds)
                        hash_s = int(line.split(":")[1].split("s")[0])
                kfb = open(kf, "rb").read() if t1 is not None else None
                return (None if t1 is None else t1 - t0), hash_s, (None if t2 is None else t2 - t0), text, kfb

            p_threads = min(cores, 64)
            want_e2e = bool(e2e and ours_exe and os.access(ours_exe, os.X_OK))
            wall, hash_s, ref_total, ref_stdout, ref_kf = run_ref(p_threads, want_e2e)
            if wall is not None:
                res = {"value": kmers / wall, "unit": "kmers/s", "cores": p_threads, "kind": "reference",
                       "sample": f"first {n} reads ({kmers} k-mers, 1/{max(n_reads_total // max(n, 1), 1)} of the workload) as FASTQ; reference "
                                 f"SOAPdenovo-Trans pregraph -K {K} -p {p_threads}, process start until "
                                 f"*.kmerFreq written ({wall:.2f} s)",
                       "hash_reads_s": hash_s,
                       "hash_reads_kmers_per_s": None if not hash_s else kmers / hash_s,
                       "kmerfreq_identical": None}
                if gpu_hist is not None:
                    # the bench's own kernels on the very reads the reference just counted: bins 1..255 as freqStat prints them
                    # (prlHashReads.c:994-1023)
                    try:
                        h = gpu_hist(n)
                        res["kmerfreq_identical"] = bool(b"".join(b"%d\n" % int(v) for v in h[1:256]) == ref_kf)
                    except Exception as ex:
                        log("kmerFreq comparison failed:", repr(ex))
                if also_p8 and p_threads != 8:
                    w8, h8, _, _, _ = run_ref(8, False)
                    res["p8"] = {"value": None if w8 is None else kmers / w8, "cores": 8, "wall_s": w8, "hash_reads_s": h8,
                                 "hash_reads_kmers_per_s": None if not h8 else kmers / h8}
                e2e_res = None
                if want_e2e and ref_total is not None:
                    e2e_res = {"reads": n, "read_len": L, "K": K, "kmers": kmers, "threads": p_threads,
                               "ref_wall_s": round(ref_total, 2),
                               "ref_cmd": f"SOAPdenovo-Trans-{variant}mer pregraph -s lib.cfg -K {K} -p {p_threads} -o ref",
                               "ref_phase_lines": [l for l in ref_stdout.splitlines() if l.startswith("time spent")]}
                    ours_out = os.path.join(tmp, "ours")
                    walls, rk = [], None
                    for _ in range(2):
                        t0 = time.time()
                        rk = subprocess.run([ours_exe, "pregraph", "-s", cfg, "-K", str(K), "-p", str(p_threads), "-o", ours_out,
                                             "--max-k", str(variant)], capture_output=True, text=True, timeout=600,
                                            env=dict(os.environ, SDT_TIMING="1"))
                        walls.append(round(time.time() - t0, 3))
                        if rk.returncode != 0:
                            break
                    e2e_res["ours_cmd"] = f"sdt-pregraph pregraph -s lib.cfg -K {K} -p {p_threads} -o ours --max-k {variant}"
                    e2e_res["ours_rc"] = rk.returncode
                    e2e_res["ours_walls_s"] = walls
                    if rk.returncode == 0:
                        e2e_res["ours_wall_s"] = max(walls)          # the slower of two runs: nothing is warmed up for the first
                        e2e_res["speedup"] = round(ref_total / max(walls), 2)
                        same = {}
                        for ext in ("kmerFreq", "vertex", "preGraphBasic", "preArc"):
                            a, b = ours_out + "." + ext, os.path.join(tmp, f"ref_p{p_threads}.{ext}")
                            same[ext] = os.path.exists(a) and os.path.exists(b) and open(a, "rb").read() == open(b, "rb").read()
                        try:
                            same["edge"] = (gzip.open(ours_out + ".edge.gz").read() ==
                                            gzip.open(os.path.join(tmp, f"ref_p{p_threads}.edge.gz")).read())
                        except OSError:
                            same["edge"] = False
                        e2e_res["identical"] = same
                        e2e_res["file_bytes"] = {ext: os.path.getsize(ours_out + "." + ext) for ext in
                                                 ("kmerFreq", "vertex", "preGraphBasic", "preArc", "edge.gz") if os.path.exists(ours_out + "." + ext)}
                        e2e_res["ours_phase_lines"] = [l for l in rk.stdout.splitlines() if l.startswith("time spent")]
                        e2e_res["ours_phase_ms"] = [l.replace("[sdt-pregraph] ", "") for l in rk.stderr.splitlines()
                                                    if l.startswith(("[sdt-pregraph]", "[device]", "[cuttip]", "[graph]", "[edges]", "[read2edge]", "[ingest]", "[libsdt_gpu]"))]
                    else:
                        e2e_res["stderr_tail"] = rk.stderr[-1500:]
                    e2e_res["note"] = ("whole pregraph stage, process start to exit, same FASTQ and config on the same box: the reference on "
                                       f"{p_threads} host threads against sdt-pregraph on one MI355X + the same host threads; five output files "
                                       "compared byte for byte (edge.gz by content)")
                return res, e2e_res
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
    ohist, _ = o.mark()
    dt = time.time() - t0
    res = {"value": kmers / dt, "unit": "kmers/s", "cores": 1, "kind": "port",
           "sample": f"first {n} reads ({kmers} k-mers), oracle/sdt_oracle.c single thread ({dt:.2f} s)", "kmerfreq_identical": None}
    if gpu_hist is not None:
        try:
            res["kmerfreq_identical"] = bool((np.asarray(gpu_hist(n), dtype=np.int64) == np.asarray(ohist, dtype=np.int64)).all())
        except Exception as ex:
            log("kmerFreq comparison failed:", repr(ex))
    return res, None


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
    ap.add_argument("--cpu-sample", type=int, default=8_000_000, help="reads timed on the CPU baseline (0 = skip); the default keeps the reference at ~25 s of wall clock on 16 cores")
    ap.add_argument("--cpu-p8", action="store_true", help="cpu_baseline: also time the reference with its default -p 8")
    ap.add_argument("--e2e", type=int, default=1, help="N=1: let the reference finish its whole pregraph on the CPU sample, run sdt-pregraph on "
                                                       "the same config and compare the five files (0 = skip: the reference is stopped at *.kmerFreq)")
    ap.add_argument("--est-distinct", type=int, default=0)
    ap.add_argument("--pipeline", choices=["auto", "direct", "superkmer"], default="auto",
                    help="pass-1 kernel family: 'direct' = one device atomic per occurrence (k_count_reads); 'superkmer' = "
                         "minimizer buckets of super-k-mers counted in LDS (k_sk_*); 'auto' = the library's default")
    ap.add_argument("--track-first", action="store_true", help="SDT_FLAG_TRACK_FIRST: what the five-file pipeline runs with")
    ap.add_argument("--extras", type=int, default=1, help="N=1: also time the TRACK_FIRST configuration and the PCIe-inclusive "
                                                          "rate through sdt_gpu_push_reads (0 = skip)")
    ap.add_argument("--force-sharded", action="store_true", help="run the N>1 code path even with one rank")
    ap.add_argument("--own-reads", action="store_true",
                    help="N > 1: rank r draws its own reads (seed 42 + 1000 r) instead of taking its slice of the single-rank workload.  "
                         "The default -- every rank generates the WHOLE workload (2 s on the device) and keeps its slice -- makes the N-rank job "
                         "count exactly the reads the 1-rank job counts: the same kmerfreq_sha1 at every N, strong scaling of ONE input")
    ap.add_argument("--slice-of-whole", action="store_true", help="(the default since round 6; accepted for old command lines)")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from soapdenovo_trans_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    dist = None
    sharded_path = world > 1 or args.force_sharded
    share = os.environ.get("SDT_BENCH_SHARE_DEVICE") == "1"
    if sharded_path:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # SDT_BENCH_SHARE_DEVICE=1 (validation only, never for a reported number): all ranks on cuda:0, so that the N>1
        # code path can be exercised on a 1-GPU box (RCCL refuses two ranks per device: shared-memory transport)
        if share:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dist.init_process_group("gloo")              # control plane only (communicator id, barrier, max time): the data path is the library's own RCCL
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    def log(*a):
        if rank == 0:
            print("[bench]", *a, file=sys.stderr, flush=True)

    K, L = pkg.clamp_K(args.K), args.read_len
    n_total = args.reads
    mode = "bucket" if sharded_path else "single"
    n_local = n_total // world + (1 if rank < n_total % world else 0)
    kmers_total = n_total * (L - K + 1)
    t0 = time.time()
    if mode == "bucket" and not args.own_reads:
        wf, _, _ = synth.torch_workload(n_total, L, args.T, dev, err=args.err, sigma=args.sigma, seed=42)
        lo = (rank * n_total // world) // 16 * 16        # slices start on a word boundary (16 reads x L bases)
        hi = ((rank + 1) * n_total // world) // 16 * 16 if rank + 1 < world else n_total
        n_local = hi - lo
        nwords = (n_local * L + 15) // 16 + 4
        words = torch.zeros(nwords, dtype=torch.int32, device=dev)
        words[: nwords - 4] = wf[lo * L // 16: lo * L // 16 + nwords - 4]
        offsets = torch.arange(n_local + 1, dtype=torch.int64, device=dev) * L
        del wf
    else:
        words, offsets, nwords = synth.torch_workload(n_local, L, args.T, dev, err=args.err, sigma=args.sigma,
                                                      seed=42 + (1000 * rank if mode == "bucket" else 0))
    torch.cuda.synchronize()
    log(f"workload: {n_local} reads x {L} bp on rank 0 ({nwords * 4 / 1e9:.2f} GB packed), generated in {time.time() - t0:.1f} s")

    # Table size from the data, not from the answer: count the nodes of up to four doubling prefixes of this rank's reads (small
    # passes outside the timed region) and extrapolate (config.distinct_nodes against config.est_distinct_per_rank shows how well).
    def estimate_distinct():
        c = min(max(n_local // 64, 1 << 18), 1 << 21, n_local // 2)
        if c < 1024:
            return n_local * (L - K + 1)
        # distinct k-mers follow a power law of the reads seen (Heaps: the transcripts saturate, the error k-mers keep coming,
        # and under deep coverage even those repeat): fit the exponent on doubling prefixes, let it drift on as it did between
        # the last two pairs, extrapolate.  (A straight line through the last two points said 1.2 G for the 678 M of the
        # 200 M-read workload; the table then has four times the slots its scans need.)
        import math
        sizes = [c]
        while len(sizes) < 4 and sizes[-1] * 4 <= n_local:
            sizes.append(sizes[-1] * 2)
        if len(sizes) < 2:
            sizes.append(2 * c)
        got = []
        for m in sizes:
            with pkg.PregraphGPU(K, est_distinct=1 << 24, device=dev.index or 0) as ge_:
                ge_.count_reads_device(words, nwords, offsets, m, L)
                got.append(max(ge_.finish_count()[1], 1))
        bs = [math.log2(got[i + 1] / got[i]) for i in range(len(got) - 1)]
        more = math.log2(n_local / sizes[-1])
        drift = max(bs[-1] - bs[-2], 0.0) if len(bs) > 1 else 0.0
        b = min(bs[-1] + drift * (more + 1) / 2, 1.0)
        est_ = int(got[-1] * 2 ** (b * more))
        log(f"distinct k-mers of the first {sizes} reads: {got}; exponents {[round(x, 3) for x in bs]} -> {b:.3f}: {est_} expected")
        return est_
    est = args.est_distinct or estimate_distinct() + (1 << 20)
    base_flags = {"auto": 0, "direct": pkg.SDT_FLAG_DIRECT, "superkmer": pkg.SDT_FLAG_PARTITION}[args.pipeline]
    flags = base_flags | (pkg.SDT_FLAG_TRACK_FIRST if args.track_first else 0)
    g = pkg.PregraphGPU(K, est_distinct=est, device=dev.index or 0, flags=flags)
    stream = torch.cuda.Stream(device=dev)
    g.set_stream(stream.cuda_stream)
    log(f"node table: {g.table_slots()} slots")

    if mode == "bucket":
        if share:
            name = [os.environ.get("MASTER_PORT", "0") + "_" + str(os.getppid())]
            dist.broadcast_object_list(name, src=0)
            g.comm_init_shm("bench" + name[0], rank, world)
        else:
            cid = [pkg.new_comm_id() if rank == 0 else None]
            dist.broadcast_object_list(cid, src=0)
            g.comm_init(cid[0], rank, world)

    def allsum(hist, kmers, nodes, linear):
        v = np.concatenate([np.asarray(hist, dtype=np.int64), np.array([kmers, nodes, linear], dtype=np.int64)])
        if mode == "bucket":
            v = g.allreduce(v)
        return v[:257], int(v[257]), int(v[258]), int(v[259])

    local_inserted = [0]

    def one_step(ctx):
        ctx.reset()
        if mode == "bucket":
            ctx.count_reads_sharded(words, nwords, offsets, n_local, L)
        else:
            ctx.count_reads_device(words, nwords, offsets, n_local, L)
        kmers, nodes = ctx.finish_count()
        local_inserted[0] = kmers              # this rank's share before the all-reduce
        if args.d:
            ctx.delow(args.d)
        hist, linear = ctx.mark_and_hist()
        hist, kmers, nodes, linear = allsum(hist, kmers, nodes, linear)
        return kmers, nodes, linear, hist

    def barrier():
        torch.cuda.synchronize()
        if sharded_path:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(ctx, steps, warmup):
        res = None
        for _ in range(warmup):
            res = one_step(ctx)
        ctx.kernel_time(reset=True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = one_step(ctx)
        barrier()
        dt = time.perf_counter() - t0
        if sharded_path:
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return res, dt

    (kmers, nodes, linear, hist), dt = timed(g, args.steps, args.warmup)
    assert kmers == kmers_total, f"processed {kmers} k-mers, expected {kmers_total}"
    assert int(hist.sum()) == nodes, "kmerFreq bins do not add up to the node count"
    stage_ms, sk_counters = g.stage_times()
    kms, launches, _ = g.kernel_time(reset=True)
    log("table slots:", g.table_slots())
    log("stage ms per step [direct, sk scatter, sk split, sk count]:", [round(x / args.steps, 2) for x in stage_ms[:4]],
        {k: v for k, v in sk_counters.items() if "ticks" not in k or v})
    ms_per_step = dt / args.steps * 1e3
    value = kmers_total * args.steps / dt
    B = algorithmic_bytes_per_kmer(L, K)
    # kernel level: the k-mers this rank chopped over the time of this rank's pass-1 kernels (N = 1: the whole job)
    local_kmers = n_local * (L - K + 1)
    roof = None
    if local_kmers and kms > 0:
        ach = B * local_kmers * args.steps / (kms * 1e-3) / 1e9
        pipeline = stage_ms[1] + stage_ms[3] > stage_ms[0]
        batches = max(sk_counters["batches"] // (args.steps + args.warmup), 1) if pipeline else max(int(launches) // args.steps, 1)
        per_launch_kmers = local_kmers / batches
        tpk, tsrc = pmc_traffic_per_kmer(K, L, n_total) if world == 1 else (None, None)
        roof = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5),
                "traffic": None if tpk is None else round(tpk * per_launch_kmers),
                "traffic_unit": "HBM bytes per launch, a LOWER bound (FETCH_SIZE x2 for the record-streaming kernels + WRITE_SIZE PMC passes of this workload, this build; FETCH_SIZE under-reports wide coalesced reads on gfx950)",
                "traffic_source": tsrc,
                "algorithmic_bytes_per_launch": round(B * per_launch_kmers),
                "kernel": ("pass 1 = k_sk_scatter_reads_seq (or k_sk_scatter_reads) + chunk lists + k_sk_scatter_records_staged + k_sk_count_flat "
                           "(every k-mer goes through all of them; a 'launch' = one batch through the pipeline)") if pipeline else "k_count_reads",
                "bytes_per_kmer": round(B, 3), "node_bytes_E": node_bytes(K),
                "E_rule": "sizeof(kmer_t) of the smallest reference variant that holds K = the key words the path computes with (31mer 24 B, 63mer 32 B, 127mer 48 B)",
                "launches": int(batches * args.steps),
                "avg_launch_ms": round(kms / (batches * args.steps), 4), "kernel_ms_per_step": round(kms / args.steps, 3),
                "stage_ms_per_step": {"scatter": round(stage_ms[1] / args.steps, 2), "split": round(stage_ms[2] / args.steps, 2),
                                      "count": round(stage_ms[3] / args.steps, 2),
                                      "direct": round(stage_ms[0] / args.steps, 2)},
                "node_table": g.table_info(),
                "merges_per_kmer": round(sk_counters["merges"] / max(local_kmers, 1), 4) if pipeline else None}
        if 31 < K <= 63:
            # SURVEY 8(d) prices its C4 (K = 63) with the node of the 127MER build (48 B: 96.3 B per k-mer); this library -- like the
            # reference's own 63mer build -- holds K = 63 in two words.  Both fractions, so that either reading can be checked:
            B127 = algorithmic_bytes_per_kmer(L, K, 127)
            roof["as_127mer_build"] = {"node_bytes_E": 48, "bytes_per_kmer": round(B127, 3), "achieved": round(ach * B127 / B, 2),
                                       "frac": round(ach * B127 / B / HBM_PEAK_GBS, 5)}
        if sharded_path:
            roof["rank"] = 0
    out = {
        "metric": "pregraph k-mers hashed/sec", "value": value, "unit": "kmers/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"{n_total} x {L} bp synthetic transcriptome reads (T={args.T}, err={args.err}), "
                               f"K={K}, pass-1 chop+hash+count" + (f"+delow(-d {args.d})" if args.d else "") + "+kmerFreq"
                               + (f", sigma={args.sigma}" if args.sigma != 2.0 else "")
                               + (", first-occurrence tracking" if args.track_first else ""), "reads": n_total, "read_len": L, "K": K,
                   "metric_definition": "HBM-resident: the packed reads are in device memory when the timed region starts; the rate from "
                                        "the first host-to-device copy to finish_count (SURVEY 8(d)) is pcie_inclusive.value, never value",
                   "kmers": kmers_total, "distinct_nodes": nodes, "linear_nodes": linear, "est_distinct_per_rank": est,
                   "parallelism": {"single": "single-GPU table",
                                   "bucket": f"reads split x{world}; tables sharded by minimizer bucket; super-k-mer chunks by grouped "
                                             f"ncclSend/ncclRecv (C ABI: sdt_gpu_count_reads_sharded)" + (" [shared-memory transport: ranks share one GPU]" if share else ""),
                                   }[mode],
                   "kmerfreq_sha1": __import__("hashlib").sha1(np.asarray(hist, dtype=np.int64).tobytes()).hexdigest()},
        "roofline": roof,
    }
    if mode == "bucket":
        sent, recv, xms, nx = g.comm_stats()
        tot = g.allreduce([sent, recv])
        # skew: the k-mers every rank counted into ITS shard in the last step (the owners' load, after the exchange)
        per_rank = [0] * world
        per_rank[rank] = int(local_inserted[0])
        per_rank = [int(x) for x in g.allreduce(per_rank)]
        out["per_rank_kmers_counted"] = per_rank
        out["skew_max_over_mean"] = round(max(per_rank) / max(sum(per_rank) / world, 1), 4)
        out["exchange"] = {"bytes_sent_rank0": sent, "bytes_sent_all_ranks": int(tot[0]), "exchanges_rank0": nx,
                           "bytes_per_kmer": round(int(tot[0]) / max(kmers_total * (args.steps + args.warmup), 1), 3),
                           "ms_on_exchange_stream_rank0": round(xms, 2),
                           "GBps_out_rank0": round(sent / max(xms, 1e-9) / 1e6, 2),
                           "GBps_per_link_rank0": round(sent / max(world - 1, 1) / max(xms, 1e-9) / 1e6, 2),
                           "note": "ms = sum over this rank's exchanges of the time between the two events around each grouped send/recv "
                                   "(payload + metas in one group); per link = bytes to one peer / that time; xGMI peak ~153 GB/s per "
                                   "link and direction"}
    # ---- extras (N = 1): the configuration the five-file pipeline runs with, and the rate from host buffers ----
    out["track_first"] = None
    out["pcie_inclusive"] = None
    if world == 1 and not sharded_path and args.extras and not args.track_first:
        g.close()
        g = None
        try:
            with pkg.PregraphGPU(K, est_distinct=est, device=dev.index or 0, flags=base_flags | pkg.SDT_FLAG_TRACK_FIRST) as gt:
                gt.set_stream(stream.cuda_stream)
                (k2, n2, _, _), dt2 = timed(gt, max(args.steps - 1, 1), 1)
                assert (k2, n2) == (kmers_total, nodes)
                st2 = max(args.steps - 1, 1)
                out["track_first"] = {"value": kmers_total * st2 / dt2, "unit": "kmers/s", "ms_per_step": dt2 / st2 * 1e3,
                                      "note": "same step with SDT_FLAG_TRACK_FIRST (first-occurrence ordinals per node: what sdt-pregraph "
                                              "needs for *.vertex / *.edge / *.preArc in the reference's order)"}
        except Exception as e:
            log("track-first extra failed:", repr(e))
        try:
            # PCIe-inclusive ("from first H2D to finish_count complete"): the first reads of the workload from PINNED
            # host memory in batches of 2^20 reads through sdt_gpu_push_reads_async (H2D into a ring of staging buffers on the
            # copy stream, kernels behind it; the host never waits for a copy), offsets built once
            nb = n_local                                 # the whole workload
            batch = 1 << 20
            hw = torch.empty((nb * L + 15) // 16 + 4, dtype=torch.int32).pin_memory()
            hw.copy_(words[: hw.numel()])
            hwn = hw.numpy().view(np.uint32)
            assert (batch * L) % 16 == 0                 # every batch starts on a word boundary
            # what the link can do: the same bytes, nothing but the copy
            dst = torch.empty_like(hw, device=dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dst.copy_(hw, non_blocking=True)
            torch.cuda.synchronize()
            h2d_s = time.perf_counter() - t0
            del dst
            # the same stream cut into RAGGED reads (alternately L - 10 and L + 10 bases: other reads, same bytes) with their
            # offsets, pinned as well: the path sdt_gpu_push_reads_async takes for reads of unequal length
            pair = np.array([L - 10, L + 10], dtype=np.uint64)
            rag_off = np.zeros(batch + 1, dtype=np.uint64)
            rag_off[1:] = np.cumsum(np.tile(pair, batch // 2))
            rag_off_t = torch.from_numpy(rag_off.view(np.int64)).pin_memory()
            rag_off_p = rag_off_t.numpy().view(np.uint64)
            rag_kmers_per_batch = int((batch // 2) * ((L - 10 - K + 1) + (L + 10 - K + 1)))
            with pkg.PregraphGPU(K, est_distinct=est, device=dev.index or 0, flags=base_flags) as gp:
                walls, rag_walls = [], []
                pcie_stage = None
                for rep in range(3):
                    gp.reset()
                    gp.finish_count()
                    gp.hint_total_kmers(nb * (L - K + 1))
                    gp.kernel_time(reset=True)
                    t0 = time.perf_counter()
                    for r0 in range(0, nb, batch):
                        nr = min(batch, nb - r0)
                        w0 = r0 * L // 16
                        gp.push_reads_fixed_async(hwn[w0: w0 + (nr * L + 15) // 16 + 4], nr, L)
                    kk, _ = gp.finish_count()
                    walls.append(time.perf_counter() - t0)
                    assert kk == nb * (L - K + 1)
                    if walls[-1] == min(walls):
                        pcie_stage = [round(x, 2) for x in gp.stage_times()[0]]
                nfull = nb // batch                      # whole batches only: every one shares the offsets array
                for rep in range(3 if nfull else 0):
                    gp.reset()
                    gp.finish_count()
                    gp.hint_total_kmers(nfull * rag_kmers_per_batch)
                    t0 = time.perf_counter()
                    for b_ in range(nfull):
                        w0 = b_ * batch * L // 16
                        gp.push_reads_async(hwn[w0: w0 + (batch * L + 15) // 16 + 4], rag_off_p)
                    kk, _ = gp.finish_count()
                    rag_walls.append(time.perf_counter() - t0)
                    assert kk == nfull * rag_kmers_per_batch
            walls.sort()
            med = walls[len(walls) // 2]
            out["pcie_inclusive"] = {"value": nb * (L - K + 1) / med, "unit": "kmers/s", "reads": nb,
                                     "frac_of_resident": round(nb * (L - K + 1) / med / value, 3),
                                     "h2d_alone_GBps": round(hw.numel() * 4 / h2d_s / 1e9, 2),
                                     "frac_of_h2d_bound": round(h2d_s / med, 3),
                                     "wall_ms": round(med * 1e3, 2), "walls_ms": [round(w * 1e3, 2) for w in walls],
                                     "stage_ms": dict(zip(("direct", "scatter", "split", "count"), pcie_stage)),
                                     "ragged": None if not rag_walls else {
                                         "value": nfull * rag_kmers_per_batch / sorted(rag_walls)[len(rag_walls) // 2], "unit": "kmers/s", "reads": nfull * batch,
                                         "walls_ms": [round(w * 1e3, 2) for w in rag_walls],
                                         "note": "the same bytes cut into reads of alternately L - 10 and L + 10 bases, pushed with their offsets "
                                                 "through sdt_gpu_push_reads_async (the path of reads of unequal length)"},
                                     "note": "the WHOLE workload from pinned host memory in batches of 2^20 reads through "
                                             "sdt_gpu_push_reads_fixed_async (first H2D -> finish_count complete; no mark/kmerFreq); value = median of "
                                             "three runs; frac_of_h2d_bound = time of the bare copy of the same bytes / time of the run: 1.0 = the link is the limit"}
        except Exception as e:
            log("pcie extra failed:", repr(e))
    if rank == 0 and world == 1 and not sharded_path and args.cpu_sample > 0:
        try:
            def gpu_hist(m):
                """257 bins of the first m reads of the workload, by the kernels the step above ran"""
                with pkg.PregraphGPU(K, est_distinct=max(m * 4, 1 << 20), device=dev.index or 0, flags=base_flags) as gh:
                    gh.count_reads_device(words, (m * L + 15) // 16 + 4, offsets, m, L)
                    gh.finish_count()
                    return gh.mark_and_hist()[0]
            out["cpu_baseline"], out["e2e"] = cpu_baseline(words, n_local, L, K, args.cpu_sample, log, also_p8=args.cpu_p8,
                                                           e2e=bool(args.e2e), gpu_hist=gpu_hist,
                                                           ours_exe=os.path.join(pkg.CSRC_DIR, "sdt-pregraph"))
        except Exception as e:     # the baseline is reported, never required
            log("cpu baseline failed:", repr(e))
            out["cpu_baseline"] = None
            out["e2e"] = None
    else:
        out["cpu_baseline"] = None
        out["e2e"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if g is not None:
        g.close()
    if sharded_path:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

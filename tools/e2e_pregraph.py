#!/usr/bin/env python3
"""End-to-end wall-clock: reference `pregraph` vs `sdt-pregraph` on the same synthetic FASTQ, all five output
files compared byte for byte (edge.gz after gunzip).  Run on the GPU box:
    python tools/e2e_pregraph.py --reads 2000000 --read-len 150 --K 31 --p 8
"""
import argparse
import gzip
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
from soapdenovo_trans_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=2_000_000)
ap.add_argument("--read-len", type=int, default=150)
ap.add_argument("--K", type=int, default=31)
ap.add_argument("--p", type=int, default=8)
ap.add_argument("--T", type=int, default=2000)
ap.add_argument("--skip-ref", action="store_true")
ap.add_argument("--sigma", type=float, default=2.0, help="log-normal sigma of the expression weights (2.5: the skew of SURVEY's C5)")
ap.add_argument("--timeout", type=int, default=300)
ap.add_argument("--no-max-rd-len", action="store_true", help="drop max_rd_len from the config (reference default: reads cut to 100)")
ap.add_argument("--cutoff", type=int, default=0, help="rd_len_cutoff for the library")
ap.add_argument("--dirty", action="store_true", help="SE layout only: N, lowercase, '.', IUPAC letters, ragged lengths, max_rd_len 20 below the read length")
ap.add_argument("--layout", choices=["se", "pe", "mixed"], default="se", help="library layout of the synthetic input")
ap.add_argument("--d", type=int, default=0, help="-d: delete k-mer links of frequency <= d")
ap.add_argument("--i", type=int, default=None, help="-i: minor-branch threshold in percent (reference default 5)")
ap.add_argument("--variant", type=int, default=0, choices=[0, 31, 63, 127], help="reference binary to compare with (default: 31 for K <= 31, else 127)")
ap.add_argument("--ref-p2", type=int, default=0, help="time the reference a second time with this -p (same FASTQ): kmerFreq must agree")
ap.add_argument("--compare-host-walks", action="store_true", help="run again with --host-walks and compare all files")
ap.add_argument("--gen-procs", type=int, default=0, help="processes that write the FASTQ (0 = one per usable CPU; fixed-width records, written in place)")
ap.add_argument("--ingest-probe", action="store_true", help="only time --hash-only in variants (as is, parse only, other thread counts) and stop")
ap.add_argument("--env-runs", default="", help="measurement: run sdt-pregraph once per variant 'name:K=V,K=V;name2:...' (environment switches of the library / CLI), report walls and phase lines, and stop")
ap.add_argument("--also-cli-args", default="", help="run sdt-pregraph once more with these extra arguments (e.g. '--gpus 4 --share-device') and compare its five files with the first run's")
ap.add_argument("--gen-only", default="", help="write the FASTQ files and lib.cfg into this directory, print the sdt-pregraph command line and stop (the directory is kept: for runs under rocprofv3, which wants the program itself after --)")
ap.add_argument("--pause", type=float, default=0.0, help="seconds to wait between the runs of --runs (the driver clears the device memory a process held: a run that starts right after another one may wait for that in hipInit / hipMalloc)")
ap.add_argument("--runs", type=int, default=1, help="run sdt-pregraph this many times (page cache, first-touch effects): the fastest is reported, all walls are listed")
args = ap.parse_args()


def _usable_cpus():
    n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except OSError:
        pass
    return n


def _fixed_records(letters, first_id, suffix=b""):
    """FASTQ records of one width: @r<10 digits><suffix> / bases / + / quality -- so that chunks can be written in place"""
    import numpy as np
    n, L = letters.shape
    W = 2 + 10 + len(suffix) + 1 + L + 3 + L + 1
    rec = np.empty((n, W), dtype=np.uint8)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
    ids = first_id + np.arange(n, dtype=np.int64)
    for d in range(10):
        rec[:, 11 - d] = (ids // 10 ** d) % 10 + 48
    o = 12
    for ch in suffix:
        rec[:, o] = ch; o += 1
    rec[:, o] = 10; o += 1
    rec[:, o:o + L] = letters; o += L
    rec[:, o] = 10; rec[:, o + 1] = ord("+"); rec[:, o + 2] = 10; o += 3
    rec[:, o:o + L] = ord("I"); o += L
    rec[:, o] = 10
    return rec.tobytes(), W


def _gen_se_chunk(a):
    path, done, n, seed0, read_len, T = a
    tx_ = synth.make_transcriptome(T, seed=42, sigma=args.sigma)
    codes, _ = synth.sample_reads(*tx_, n_reads=n, read_len=read_len, seed=seed0 + done, err=0.002)
    blob, W = _fixed_records(synth.BASES[codes].reshape(n, read_len), done)
    fd = os.open(path, os.O_WRONLY)
    os.pwrite(fd, blob, done * W)
    os.close(fd)
    return n


def _gen_pe_chunk(a):
    p1, p2, done, n, seed0, read_len, T = a
    tx_ = synth.make_transcriptome(T, seed=42, sigma=args.sigma)
    (c1, _), (c2, _) = synth.sample_pairs(*tx_, n_pairs=n, read_len=read_len, seed=seed0 + done, err=0.002, avg_ins=300)
    for path, c, suf in ((p1, c1, b"/1"), (p2, c2, b"/2")):
        blob, W = _fixed_records(synth.BASES[c].reshape(n, read_len), done, suf)
        fd = os.open(path, os.O_WRONLY)
        os.pwrite(fd, blob, done * W)
        os.close(fd)
    return n


tmp = tempfile.mkdtemp(prefix="sdt_e2e_") if not args.gen_only else (os.makedirs(args.gen_only, exist_ok=True) or args.gen_only)
try:
    tx = synth.make_transcriptome(args.T, seed=42, sigma=args.sigma)
    fq = os.path.join(tmp, "reads.fq")
    t0 = time.time()

    def write_se(path, n_reads, seed0):
        if not args.dirty:                                   # fixed-width records, chunks written in place by a pool of processes
            import multiprocessing as mp
            open(path, "wb").close()
            jobs = [(path, d, min(250_000, n_reads - d), seed0, args.read_len, args.T) for d in range(0, n_reads, 250_000)]
            with mp.Pool(args.gen_procs or _usable_cpus()) as pool:
                assert sum(pool.imap_unordered(_gen_se_chunk, jobs, chunksize=1)) == n_reads
            if os.path.getsize(path) % 32768 == 0:
                open(path, "ab").write(b"\n")
            return
        with open(path, "wb") as fo:
            done = 0
            while done < n_reads:
                n = min(250_000, n_reads - done)
                codes, offs = synth.sample_reads(*tx, n_reads=n, read_len=args.read_len, seed=seed0 + done, err=0.002)
                letters = synth.BASES[codes].reshape(n, args.read_len)
                qual = b"I" * args.read_len
                if args.dirty:
                    import numpy as np
                    rng = np.random.default_rng(seed0 + done)
                    letters = letters.copy()
                    u = rng.random(letters.shape)
                    letters[u < 0.004] = ord("N")
                    letters[(u >= 0.004) & (u < 0.006)] = ord(".")
                    low = (u >= 0.006) & (u < 0.03)
                    letters[low] = letters[low] + 32
                    letters[(u >= 0.03) & (u < 0.031)] = ord("n")
                    letters[(u >= 0.031) & (u < 0.032)] = ord("R")
                    lens = rng.integers(args.K - 5, args.read_len + 1, size=n)
                    fo.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (done + i, letters[i, :lens[i]].tobytes(), qual[:lens[i]]) for i in range(n)))
                else:
                    fo.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (done + i, letters[i].tobytes(), qual) for i in range(n)))
                done += n
        if os.path.getsize(path) % 32768 == 0:
            open(path, "ab").write(b"\n")

    def write_pe(p1, p2, n_pairs, seed0):
        qual = b"I" * args.read_len
        if True:
            import multiprocessing as mp
            open(p1, "wb").close(); open(p2, "wb").close()
            jobs = [(p1, p2, d, min(250_000, n_pairs - d), seed0, args.read_len, args.T) for d in range(0, n_pairs, 250_000)]
            with mp.Pool(args.gen_procs or _usable_cpus()) as pool:
                assert sum(pool.imap_unordered(_gen_pe_chunk, jobs, chunksize=1)) == n_pairs
            for f in (p1, p2):
                if os.path.getsize(f) % 32768 == 0:
                    open(f, "ab").write(b"\n")
            return
        with open(p1, "wb") as o1, open(p2, "wb") as o2:
            done = 0
            while done < n_pairs:
                n = min(250_000, n_pairs - done)
                (c1, _), (c2, _) = synth.sample_pairs(*tx, n_pairs=n, read_len=args.read_len, seed=seed0 + done, err=0.002, avg_ins=300)
                l1, l2 = synth.BASES[c1].reshape(n, args.read_len), synth.BASES[c2].reshape(n, args.read_len)
                o1.write(b"".join(b"@r%d/1\n%s\n+\n%s\n" % (done + i, l1[i].tobytes(), qual) for i in range(n)))
                o2.write(b"".join(b"@r%d/2\n%s\n+\n%s\n" % (done + i, l2[i].tobytes(), qual) for i in range(n)))
                done += n
        for f in (p1, p2):
            if os.path.getsize(f) % 32768 == 0:
                open(f, "ab").write(b"\n")

    cfg_path = os.path.join(tmp, "lib.cfg")
    if args.layout == "se":
        write_se(fq, args.reads, 1000)
        synth.write_config(cfg_path, args.read_len - (20 if args.dirty else 0), fastq=[fq])
    elif args.layout == "pe":
        p1, p2 = os.path.join(tmp, "r_1.fq"), os.path.join(tmp, "r_2.fq")
        write_pe(p1, p2, args.reads // 2, 1000)
        fq = p1
        open(cfg_path, "w").write(f"max_rd_len={args.read_len}\n[LIB]\navg_ins=300\nreverse_seq=0\nasm_flags=3\nq1={p1}\nq2={p2}\n")
    else:
        # mixed: the config lists a 500-bp single-end library FIRST and a 200-bp paired one second, with a second pair of
        # files and a reverse_seq library; the reference sorts libraries by avg_ins and reads pairs before singles
        a1, a2 = os.path.join(tmp, "a_1.fq"), os.path.join(tmp, "a_2.fq")
        b1, b2 = os.path.join(tmp, "b_1.fq"), os.path.join(tmp, "b_2.fq")
        s1, s2 = os.path.join(tmp, "s1.fq"), os.path.join(tmp, "s2.fq")
        q = args.reads // 8
        write_pe(a1, a2, q, 1000)
        write_pe(b1, b2, q, 5000000)
        write_se(s1, 2 * q, 9000000)
        write_se(s2, args.reads - 6 * q, 13000000)
        fq = a1
        open(cfg_path, "w").write(
            f"max_rd_len={args.read_len}\n[LIB]\navg_ins=500\nreverse_seq=1\nasm_flags=3\nq={s1}\n"
            f"[LIB]\navg_ins=200\nreverse_seq=0\nasm_flags=3\nq1={a1}\nq2={a2}\nq1={b1}\nq2={b2}\nq={s2}\n"
            f"[LIB]\navg_ins=300\nasm_flags=2\nq={s1}\n")
    if args.no_max_rd_len or args.cutoff:
        txt = open(cfg_path).read()
        if args.no_max_rd_len:
            txt = "\n".join(l for l in txt.splitlines() if not l.startswith("max_rd_len")) + "\n"
        if args.cutoff:
            txt = txt.replace("[LIB]\n", "[LIB]\nrd_len_cutoff=%d\n" % args.cutoff)
        open(cfg_path, "w").write(txt)
    gen_s = time.time() - t0
    res = {"reads": args.reads, "read_len": args.read_len, "K": args.K, "p": args.p, "layout": args.layout, "fastq_bytes": os.path.getsize(fq),
           "kmers": args.reads * (args.read_len - args.K + 1), "gen_s": round(gen_s, 1)}
    subprocess.run("cat %s/*.fq > /dev/null" % tmp, shell=True)                 # warm the page cache
    ours = os.path.join(pkg.CSRC_DIR, "sdt-pregraph")
    variant = args.variant or (31 if args.K <= 31 else 127)
    common = (["-d", str(args.d)] if args.d else []) + (["-i", str(args.i)] if args.i is not None else [])
    extra = ["--max-k", str(variant)] + common
    res["variant"], res["d"] = variant, args.d
    os.environ["SDT_TIMING"] = "1"
    if args.gen_only:
        print(" ".join([ours, "pregraph", "-s", os.path.join(tmp, "lib.cfg"), "-K", str(args.K), "-p", str(args.p), "-o", os.path.join(tmp, "ours")] + extra))
        sys.stdout.flush()
        os._exit(0)                                            # (keeps the directory: the finally below removes it otherwise)
    if args.ingest_probe:
        probe = {}
        for name, env, p_ in (("as_is", {}, args.p), ("as_is_again", {}, args.p), ("parse_only", {"SDT_PARSE_ONLY": "1"}, args.p),
                              ("t8", {"SDT_PARSE_THREADS": "8"}, args.p), ("t12", {"SDT_PARSE_THREADS": "12"}, args.p),
                              ("t12_block", {"SDT_PARSE_THREADS": "12", "SDT_SYNC": "block"}, args.p), ("t12_block2", {"SDT_PARSE_THREADS": "12", "SDT_SYNC": "block"}, args.p),
                              ("t12_nopool", {"SDT_PARSE_THREADS": "12", "SDT_NO_PINNED_POOL": "1"}, args.p)):
            t0 = time.time()
            rp = subprocess.run([ours, "pregraph", "-s", os.path.join(tmp, "lib.cfg"), "-K", str(args.K), "-p", str(p_), "-o", os.path.join(tmp, "probe"),
                                 "--hash-only"] + extra, capture_output=True, text=True, env=dict(os.environ, **env), timeout=120)
            probe[name] = {"wall_s": round(time.time() - t0, 2), "phases": [l.replace("[sdt-pregraph] ", "") for l in rp.stderr.splitlines() if "parse + hash" in l or l.startswith("[ingest]")]}
        res["ingest_probe"] = probe
        print(json.dumps(res, indent=1))
        raise SystemExit(0)
    if args.env_runs:
        out = {}
        for spec in args.env_runs.split(";"):
            name, _, kv = spec.partition(":")
            env = dict(x.split("=", 1) for x in kv.split(",") if x)
            t0 = time.time()
            rp = subprocess.run([ours, "pregraph", "-s", os.path.join(tmp, "lib.cfg"), "-K", str(args.K), "-p", str(args.p), "-o", os.path.join(tmp, "probe")] + extra,
                                capture_output=True, text=True, env=dict(os.environ, **env), timeout=args.timeout)
            out[name] = {"env": env, "rc": rp.returncode, "wall_s": round(time.time() - t0, 2),
                         "phases": [l.replace("[sdt-pregraph] ", "") for l in rp.stderr.splitlines() if l.startswith(("[sdt-pregraph]", "[ingest]", "[device]", "[libsdt_gpu]"))]}
        res["env_runs"] = out
        print(json.dumps(res, indent=1))
        raise SystemExit(0)
    walls, r = [], None
    for _run in range(max(1, args.runs)):
        if _run and args.pause > 0:
            time.sleep(args.pause)
        t0 = time.time()
        try:
            rk = subprocess.run([ours, "pregraph", "-s", os.path.join(tmp, "lib.cfg"), "-K", str(args.K), "-p", str(args.p), "-o",
                                 os.path.join(tmp, "ours")] + extra, capture_output=True, text=True, timeout=args.timeout)
        except subprocess.TimeoutExpired as e:
            print("sdt-pregraph timed out; stderr so far:\n", (e.stderr or b"").decode()[-3000:])
            raise SystemExit(1)
        w = round(time.time() - t0, 2)
        if rk.returncode != 0:
            print(rk.stdout[-2000:], rk.stderr[-2000:])
            raise SystemExit("sdt-pregraph failed")
        if not walls or w < min(walls):
            r = rk
        walls.append(w)
    res["ours_wall_s"] = min(walls)
    res["ours_walls_s"] = walls
    t0 = time.time()
    r2 = subprocess.run([ours, "pregraph", "-s", os.path.join(tmp, "lib.cfg"), "-K", str(args.K), "-p", str(args.p), "-o",
                         os.path.join(tmp, "ours_hash"), "--hash-only"] + extra, capture_output=True, text=True, timeout=args.timeout)
    res["ours_hash_only_wall_s"] = round(time.time() - t0, 2)
    if args.also_cli_args:
        t0 = time.time()
        ra = subprocess.run([ours, "pregraph", "-s", os.path.join(tmp, "lib.cfg"), "-K", str(args.K), "-p", str(args.p), "-o",
                             os.path.join(tmp, "also")] + extra + args.also_cli_args.split(), capture_output=True, text=True, timeout=args.timeout)
        res["also"] = {"cli_args": args.also_cli_args, "rc": ra.returncode, "wall_s": round(time.time() - t0, 2),
                       "phase_ms": [l.replace("[sdt-pregraph] ", "") for l in ra.stderr.splitlines() if l.startswith(("[sdt-pregraph]", "[read2edge]"))],
                       "same_as_first_run": {ext: (os.path.exists(os.path.join(tmp, "also." + ext)) and
                                                   open(os.path.join(tmp, "ours." + ext), "rb").read() == open(os.path.join(tmp, "also." + ext), "rb").read())
                                             for ext in ("kmerFreq", "vertex", "preGraphBasic", "preArc", "edge.gz")}}
        if ra.returncode != 0:
            res["also"]["stderr_tail"] = ra.stderr[-1500:]
    if args.compare_host_walks:
        t0 = time.time()
        r3 = subprocess.run([ours, "pregraph", "-s", os.path.join(tmp, "lib.cfg"), "-K", str(args.K), "-p", str(args.p), "-o",
                             os.path.join(tmp, "hw"), "--host-walks"] + extra, capture_output=True, text=True, timeout=args.timeout)
        res["ours_host_walks_wall_s"] = round(time.time() - t0, 2)
        res["host_walks_phase_ms"] = [l.replace("[sdt-pregraph] ", "") for l in r3.stderr.splitlines() if l.startswith("[sdt-pregraph]")]
        res["same_as_host_walks"] = {ext: open(os.path.join(tmp, "ours." + ext), "rb").read() == open(os.path.join(tmp, "hw." + ext), "rb").read()
                                     for ext in ("kmerFreq", "vertex", "preGraphBasic", "preArc", "edge.gz")}
        strip = lambda t: [l for l in t.splitlines() if not l.startswith("time spent")]
        res["same_stdout_as_host_walks"] = strip(r.stdout) == strip(r3.stdout)
    if not args.skip_ref:
        ref = os.path.join(ROOT, "oracle", "_ref", f"SOAPdenovo-Trans-{variant}mer")
        t0 = time.time()
        try:
            rr = subprocess.run([ref, "pregraph", "-s", os.path.join(tmp, "lib.cfg"), "-K", str(args.K), "-p", str(args.p), "-o",
                                 os.path.join(tmp, "ref")] + common, capture_output=True, text=True, timeout=max(args.timeout, 900))
        except subprocess.TimeoutExpired:
            print(json.dumps(res, indent=1))
            raise SystemExit("the reference binary did not finish (its AIO reader can spin forever, SURVEY 9.3-q9)")
        res["ref_wall_s"] = round(time.time() - t0, 2)
        res["ref_phase_lines"] = [l for l in rr.stdout.splitlines() if l.startswith("time spent")]
        same = {}
        for ext in ("kmerFreq", "vertex", "preGraphBasic", "preArc"):
            same[ext] = open(os.path.join(tmp, "ours." + ext), "rb").read() == open(os.path.join(tmp, "ref." + ext), "rb").read()
        same["edge"] = gzip.open(os.path.join(tmp, "ours.edge.gz")).read() == gzip.open(os.path.join(tmp, "ref.edge.gz")).read()
        res["identical"] = same
        if not same["preArc"]:
            a = open(os.path.join(tmp, "ours.preArc")).read().splitlines()
            b = open(os.path.join(tmp, "ref.preArc")).read().splitlines()
            res["preArc_lines"] = [len(a), len(b)]
            res["preArc_diff"] = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y][:6]
            rh = subprocess.run([ours, "pregraph", "-s", os.path.join(tmp, "lib.cfg"), "-K", str(args.K), "-p", str(args.p), "-o",
                                 os.path.join(tmp, "hm"), "--host-map"] + extra, capture_output=True, text=True, timeout=args.timeout)
            res["host_map_preArc_same_as_ref"] = open(os.path.join(tmp, "hm.preArc")).read() == open(os.path.join(tmp, "ref.preArc")).read()
        res["speedup_full"] = round(res["ref_wall_s"] / res["ours_wall_s"], 2)
        hr = [l for l in rr.stdout.splitlines() if l.startswith("time spent on hash reads")]
        if hr:                                       # the reference's own line (prlHashReads.c:623): parse + chop + insert
            res["ref_hash_reads_s"] = int(hr[0].split(":")[1].split("s")[0])
            res["ref_hash_reads_kmers_per_s"] = round(res["kmers"] / max(res["ref_hash_reads_s"], 1))
        if args.ref_p2:
            t0 = time.time()
            rr2 = subprocess.run([ref, "pregraph", "-s", os.path.join(tmp, "lib.cfg"), "-K", str(args.K), "-p", str(args.ref_p2), "-o",
                                  os.path.join(tmp, "ref2")] + common, capture_output=True, text=True, timeout=max(args.timeout, 1800))
            res["ref_p2"] = args.ref_p2
            res["ref_p2_wall_s"] = round(time.time() - t0, 2)
            res["ref_p2_phase_lines"] = [l for l in rr2.stdout.splitlines() if l.startswith("time spent")]
            res["ref_p2_kmerFreq_same"] = open(os.path.join(tmp, "ref2.kmerFreq"), "rb").read() == open(os.path.join(tmp, "ref.kmerFreq"), "rb").read()
            res["speedup_full_vs_p2"] = round(res["ref_p2_wall_s"] / res["ours_wall_s"], 2)
    res["ours_phase_lines"] = [l for l in r.stdout.splitlines() if l.startswith("time spent")]
    res["ours_phase_ms"] = [l.replace("[sdt-pregraph] ", "") for l in r.stderr.splitlines() if l.startswith(("[sdt-pregraph]", "[cuttip]", "[graph]", "[edges]", "[read2edge]", "[ingest]", "[device]"))]
    print(json.dumps(res, indent=1))
finally:
    shutil.rmtree(tmp, ignore_errors=True)

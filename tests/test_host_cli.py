"""The C host (soapdenovo-trans_amd/csrc/host): config parser + parallel FASTQ/FASTA reader + 2-bit packer
(CPU, via sdt-readdump) and the sdt-pregraph command line end to end on the golden cases (GPU)."""
import gzip
import os
import re
import subprocess

import numpy as np
import pytest

import golden_util as gu


def materialise(info, tmp):
    d = info["dir"]
    for f in os.listdir(d):
        if f.endswith(".fq.gz"):
            with gzip.open(os.path.join(d, f), "rb") as fi, open(os.path.join(tmp, f[:-3]), "wb") as fo:
                fo.write(fi.read())
    cfg = os.path.join(tmp, "lib.cfg")
    with open(os.path.join(d, "lib.cfg.template")) as fi, open(cfg, "w") as fo:
        fo.write(fi.read().replace("@DIR@", str(tmp)))
    return cfg


def bin_path(pkg, name):
    p = os.path.join(pkg.CSRC_DIR, name)
    if not os.path.exists(p):
        pkg.build()
    return p


@pytest.mark.parametrize("name", gu.case_names())
@pytest.mark.parametrize("threads,chunk", [(1, 1 << 30), (4, 70000)])
def test_reader_matches_reference_ingest(pkg, tmp_path, name, threads, chunk):
    """every read the host hands to the GPU == readseqfq's coding/truncation (oracle restatement), for one big
    chunk and for many small chunks cut at record boundaries by 4 threads"""
    info = gu.load_case(name)
    cfg = materialise(info, tmp_path)
    out = subprocess.run([bin_path(pkg, "sdt-readdump"), cfg, str(threads), str(chunk)], check=True,
                         capture_output=True, text=True).stdout.splitlines()
    got = [l for l in out if not l.startswith("#")]
    codes, offs = gu.case_reads(info)
    letters = np.frombuffer(b"ACTG", dtype=np.uint8)[codes].tobytes().decode()
    o = offs.astype(np.int64)
    want = [letters[o[i]:o[i + 1]] for i in range(len(o) - 1)]
    if info["kind"] == "pe":          # the host reads file 1 then file 2; the reference interleaves: same multiset
        assert got[: len(got) // 2] == want[0::2] and got[len(got) // 2:] == want[1::2]
    else:
        assert got == want


def test_config_parser(pkg, tmp_path):
    f = tmp_path / "a.fa"
    f.write_text(">x\nACGTNN..acgt\n>y\nTTTT\nGGGG\n")
    cfg = tmp_path / "c.cfg"
    cfg.write_text(f"#comment\nmax_rd_len=9\n[LIB]\navg_ins=500\nasm_flags=2\nf={f}\n[LIB]\navg_ins=200\nreverse_seq=1\n"
                   f"rd_len_cutoff=6\nf={f}\n[LIB]\navg_ins = 100\nasm_flags=1\nf={f}\n")
    out = subprocess.run([bin_path(pkg, "sdt-readdump"), str(cfg), "2"], check=True, capture_output=True,
                         text=True).stdout.splitlines()
    assert out[0] == "#libs 3 max_rd_len 9"
    libs = [l for l in out if l.startswith("#lib ")]
    # sorted by avg_ins (lib.c:437); "avg_ins = 100" is not recognised (blanks belong to the token) -> avg_ins 0
    assert libs == ["#lib avg_ins 0 asm_flag 1 reverse 0 rd_len_cutoff 0",
                    "#lib avg_ins 200 asm_flag 3 reverse 1 rd_len_cutoff 6",
                    "#lib avg_ins 500 asm_flag 2 reverse 0 rd_len_cutoff 0"]
    reads = [l for l in out if not l.startswith("#")]
    # lib 1: cut to 9 chars "ACGTNN..a" -> A C G T G G A A A ; second record is multi-line -> concatenated (8 chars)
    # lib 2: cut to 6 then reverse-complemented; lib 3 (asm_flags=2) is not read by pregraph
    assert reads[:2] == ["ACGTGGAAA", "TTTTGGGG"]
    assert reads[2:] == ["CCACGT", "CCAAAA"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", gu.case_names())
def test_cli_kmerfreq_bit_identical(pkg, tmp_path, name):
    info = gu.load_case(name)
    cfg = materialise(info, tmp_path)
    cmd = [bin_path(pkg, "sdt-pregraph"), "pregraph", "-s", cfg, "-K", str(info["K"]), "-p", "3", "-o",
           str(tmp_path / "out"), "--max-k", str(gu.VARIANT_MAXK[info["variant"]])]
    if info["d"]:
        cmd += ["-d", str(info["d"])]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert open(tmp_path / "out.kmerFreq").read() == gu.golden_text(info, "kmerFreq")
    m = re.search(r"(\d+) nodes allocated, (\d+) kmer in reads, (\d+) kmer processed", r.stdout)
    assert (int(m.group(1)), int(m.group(2))) == (info["nodes_allocated"], info["kmer_in_reads"])
    assert int(re.search(r"(\d+) linear nodes", r.stdout).group(1)) == info["linear_nodes"]
    if info["d"]:
        assert int(re.search(r"(\d+) kmer removed", r.stdout).group(1)) == info["kmer_removed"]


@pytest.mark.gpu
def test_cli_usage_and_errors(pkg, tmp_path):
    exe = bin_path(pkg, "sdt-pregraph")
    r = subprocess.run([exe, "pregraph"], capture_output=True, text=True)
    assert r.returncode != 0 and "pregraph -s configFile -o outputGraph" in r.stdout
    r = subprocess.run([exe, "-s", str(tmp_path / "missing.cfg"), "-o", str(tmp_path / "o")], capture_output=True,
                       text=True)
    assert r.returncode != 0 and "Cannot open" in r.stdout

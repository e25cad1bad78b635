#!/usr/bin/env python3
"""write a synthetic single-end FASTQ + library config: gen_fastq.py <dir> <reads> [read_len] [T]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.load_package()
from soapdenovo_trans_amd import synth
d, n = sys.argv[1], int(sys.argv[2])
L = int(sys.argv[3]) if len(sys.argv) > 3 else 150
T = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
os.makedirs(d, exist_ok=True)
tx = synth.make_transcriptome(T, seed=42)
fq = os.path.join(d, "reads.fq")
with open(fq, "wb") as fo:
    done = 0
    while done < n:
        m = min(250_000, n - done)
        codes, offs = synth.sample_reads(*tx, n_reads=m, read_len=L, seed=1000 + done, err=0.002)
        letters = synth.BASES[codes].reshape(m, L)
        q = b"I" * L
        fo.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (done + i, letters[i].tobytes(), q) for i in range(m)))
        done += m
if os.path.getsize(fq) % 32768 == 0:
    open(fq, "ab").write(b"\n")
synth.write_config(os.path.join(d, "lib.cfg"), L, fastq=[fq])
print(os.path.join(d, "lib.cfg"))

#!/usr/bin/env python3
"""write synthetic paired FASTQ files + library config: gen_fastq_pairs.py <dir> <pairs> [read_len] [T] [avg_ins]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.load_package()
from soapdenovo_trans_amd import synth
d, n = sys.argv[1], int(sys.argv[2])
L = int(sys.argv[3]) if len(sys.argv) > 3 else 150
T = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
ins = int(sys.argv[5]) if len(sys.argv) > 5 else 300
os.makedirs(d, exist_ok=True)
tx = synth.make_transcriptome(T, seed=42)
f1, f2 = os.path.join(d, "r_1.fq"), os.path.join(d, "r_2.fq")
q = b"I" * L
with open(f1, "wb") as o1, open(f2, "wb") as o2:
    done = 0
    while done < n:
        m = min(250_000, n - done)
        (c1, _), (c2, _) = synth.sample_pairs(*tx, n_pairs=m, read_len=L, seed=1000 + done, err=0.002, avg_ins=ins)
        l1, l2 = synth.BASES[c1].reshape(m, L), synth.BASES[c2].reshape(m, L)
        o1.write(b"".join(b"@r%d/1\n%s\n+\n%s\n" % (done + i, l1[i].tobytes(), q) for i in range(m)))
        o2.write(b"".join(b"@r%d/2\n%s\n+\n%s\n" % (done + i, l2[i].tobytes(), q) for i in range(m)))
        done += m
for f in (f1, f2):
    if os.path.getsize(f) % 32768 == 0:
        open(f, "ab").write(b"\n")
with open(os.path.join(d, "lib.cfg"), "w") as fo:
    fo.write(f"max_rd_len={L}\n[LIB]\navg_ins={ins}\nreverse_seq=0\nasm_flags=3\nq1={f1}\nq2={f2}\n")
print(os.path.join(d, "lib.cfg"))

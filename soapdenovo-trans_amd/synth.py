"""Deterministic synthetic transcriptome reads and the packed 2-bit read format.

Transcriptome: T sequences of i.i.d. uniform ACGT, length U[500, 4000]; expression weights
log-normal(0, sigma) x length; reads sampled uniformly along the chosen transcript, strand flipped
w.p. 0.5, substitution error rate e per base, no N, quality all 'I', names @r<i>.
Base coding is the reference's (inc/def.h:39-42): A=0 C=1 T=2 G=3, complement = code ^ 2.

numpy path  : small inputs for tests / goldens / the C host (also writes FASTQ).
torch path  : the bench workload generated directly in HBM (same distribution, different RNG stream).
"""
from __future__ import annotations

import os

import numpy as np

BASES = np.frombuffer(b"ACTG", dtype=np.uint8)   # int2base (inc/def.h:40)


def make_transcriptome(T: int, seed: int = 42, lo: int = 500, hi: int = 4000, sigma: float = 2.0):
    """Returns (codes uint8[total], starts int64[T+1], weights float64[T])."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(lo, hi + 1, size=T)
    starts = np.zeros(T + 1, dtype=np.int64)
    np.cumsum(lens, out=starts[1:])
    codes = rng.integers(0, 4, size=int(starts[-1]), dtype=np.uint8)
    w = rng.lognormal(0.0, sigma, size=T) * lens
    return codes, starts, w / w.sum()


def sample_reads(codes, starts, weights, n_reads: int, read_len: int, seed: int = 1, err: float = 0.002,
                 ragged: bool = False):
    """Single-end reads. Returns (read codes uint8[sum len], offsets uint64[n+1])."""
    rng = np.random.default_rng(seed)
    T = len(weights)
    lens_t = (starts[1:] - starts[:-1])
    t = rng.choice(T, size=n_reads, p=weights)
    if ragged:
        rl = rng.integers(max(8, read_len // 4), read_len + 1, size=n_reads)
    else:
        rl = np.full(n_reads, read_len, dtype=np.int64)
    rl = np.minimum(rl, lens_t[t])
    pos = (rng.random(n_reads) * (lens_t[t] - rl + 1)).astype(np.int64) + starts[t]
    flip = rng.random(n_reads) < 0.5
    offsets = np.zeros(n_reads + 1, dtype=np.uint64)
    np.cumsum(rl, out=offsets[1:])
    total = int(offsets[-1])
    # gather
    rid = np.repeat(np.arange(n_reads), rl)
    within = np.arange(total) - np.repeat(offsets[:-1].astype(np.int64), rl)
    f = flip[rid]
    src = np.where(f, pos[rid] + (rl[rid] - 1 - within), pos[rid] + within)
    out = codes[src]
    out = np.where(f, out ^ 2, out).astype(np.uint8)
    if err > 0:
        e = rng.random(total) < err
        out[e] = (out[e] + rng.integers(1, 4, size=int(e.sum()), dtype=np.uint8)) & 3
    return out, offsets


def sample_pairs(codes, starts, weights, n_pairs: int, read_len: int, seed: int = 1, err: float = 0.002,
                 avg_ins: int = 200):
    """Paired-end: mate1 = fragment[:L] forward, mate2 = revcomp(fragment)[:L]; fragment ~ N(avg_ins, 10%).
    Returns ((codes1, offsets1), (codes2, offsets2))."""
    rng = np.random.default_rng(seed)
    T = len(weights)
    lens_t = (starts[1:] - starts[:-1])
    t = rng.choice(T, size=n_pairs, p=weights)
    frag = np.clip(rng.normal(avg_ins, 0.1 * avg_ins, size=n_pairs).astype(np.int64), read_len, None)
    frag = np.minimum(frag, lens_t[t])
    rl = np.minimum(read_len, frag)
    pos = (rng.random(n_pairs) * (lens_t[t] - frag + 1)).astype(np.int64) + starts[t]
    flip = rng.random(n_pairs) < 0.5           # which strand the fragment came from
    outs = []
    for mate in (0, 1):
        offsets = np.zeros(n_pairs + 1, dtype=np.uint64)
        np.cumsum(rl, out=offsets[1:])
        total = int(offsets[-1])
        rid = np.repeat(np.arange(n_pairs), rl)
        within = np.arange(total) - np.repeat(offsets[:-1].astype(np.int64), rl)
        rev = flip[rid] ^ (mate == 1)
        src = np.where(rev, pos[rid] + (frag[rid] - 1 - within), pos[rid] + within)
        out = codes[src]
        out = np.where(rev, out ^ 2, out).astype(np.uint8)
        if err > 0:
            e = rng.random(total) < err
            out[e] = (out[e] + rng.integers(1, 4, size=int(e.sum()), dtype=np.uint8)) & 3
        outs.append((out, offsets))
    return outs[0], outs[1]


def pack_2bit(read_codes: np.ndarray, pad_words: int = 4) -> np.ndarray:
    """codes (one per byte, values 0..3) -> uint32 words, 16 bases per word, first base in bits 31..30
    (include/sdt_gpu.h 'Packed reads'), zero padded by pad_words words."""
    n = read_codes.size
    nw = (n + 15) // 16
    buf = np.zeros(nw * 16, dtype=np.uint32)
    buf[:n] = read_codes
    buf = buf.reshape(nw, 16)
    shifts = (30 - 2 * np.arange(16)).astype(np.uint32)
    words = np.bitwise_or.reduce(buf << shifts, axis=1).astype(np.uint32)
    return np.concatenate([words, np.zeros(pad_words, dtype=np.uint32)])


def write_fastq(path: str, read_codes: np.ndarray, offsets: np.ndarray, name_prefix: str = "r") -> int:
    """4-line FASTQ.  Avoids file sizes that are exact multiples of 32768 bytes: the reference's AIORead
    treats a full 32 KiB chunk as 'not the last' and then spins forever (survey 9.3-q9)."""
    letters = BASES[read_codes]
    offs = offsets.astype(np.int64)
    parts = []
    for i in range(len(offs) - 1):
        s = letters[offs[i]:offs[i + 1]].tobytes()
        parts.append(b"@" + name_prefix.encode() + str(i).encode() + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n")
    blob = b"".join(parts)
    if len(blob) % 32768 == 0 and parts:
        # lengthen the last read's name by one character
        last = parts[-1]
        parts[-1] = last[:1] + b"x" + last[1:]
        blob = b"".join(parts)
    with open(path, "wb") as fo:
        fo.write(blob)
    return len(blob)


def write_config(path: str, max_rd_len: int, fastq=None, fastq_pairs=None, avg_ins: int = 200,
                 asm_flags: int = 3, reverse_seq: int = 0, extra: str = "") -> None:
    """Library config in the reference's format (lib.c:118-438; README.md:117-147)."""
    with open(path, "w") as fo:
        fo.write(f"max_rd_len={max_rd_len}\n[LIB]\navg_ins={avg_ins}\nreverse_seq={reverse_seq}\nasm_flags={asm_flags}\n")
        fo.write(extra)
        for a, b in (fastq_pairs or []):
            fo.write(f"q1={os.path.abspath(a)}\nq2={os.path.abspath(b)}\n")
        for q in (fastq or []):
            fo.write(f"q={os.path.abspath(q)}\n")


# ------------------------------------------------------------------------------------------------ torch (device) generator

def torch_workload(n_reads: int, read_len: int, T: int, device, seed: int = 42, err: float = 0.002,
                   sigma: float = 2.0, chunk: int = 1 << 20, tx_seed: int = 42):
    """Generate the bench workload directly in device memory (tx_seed fixes the transcriptome, seed the
    read sampling: ranks share the former and differ in the latter).

    Returns (words int32[nwords] (bit pattern of the uint32 packed stream, 4 pad words),
             offsets int64[n_reads+1], nwords).  Reads are fixed length (Illumina-like), so read i starts
    at base i*read_len."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    codes_np, starts_np, w_np = make_transcriptome(T, seed=tx_seed, sigma=sigma)
    tx = torch.from_numpy(codes_np).to(device)
    starts = torch.from_numpy(starts_np).to(device)
    lens_t = starts[1:] - starts[:-1]
    weights = torch.from_numpy(w_np).to(device=device, dtype=torch.float32)
    cdf = torch.cumsum(weights.double(), 0)
    cdf = cdf / cdf[-1]
    total_bases = n_reads * read_len
    nwords = (total_bases + 15) // 16 + 4
    words = torch.zeros(nwords, dtype=torch.int32, device=device)
    # chunks must start on a word boundary: chunk*read_len % 16 == 0
    chunk = max(16, chunk // 16 * 16)
    shifts = (30 - 2 * torch.arange(16, device=device, dtype=torch.int64))
    ar = torch.arange(read_len, device=device, dtype=torch.int64)
    for c0 in range(0, n_reads, chunk):
        n = min(chunk, n_reads - c0)
        u = torch.rand(n, generator=g, device=device, dtype=torch.float64)
        t = torch.searchsorted(cdf, u).clamp_(max=T - 1)
        span = (lens_t[t] - read_len + 1).clamp_(min=1)
        pos = (torch.rand(n, generator=g, device=device, dtype=torch.float64) * span).long() + starts[t]
        flip = torch.rand(n, generator=g, device=device) < 0.5
        idx = torch.where(flip[:, None], pos[:, None] + (read_len - 1 - ar)[None, :], pos[:, None] + ar[None, :])
        rc = tx[idx]
        rc = torch.where(flip[:, None], rc ^ 2, rc)
        if err > 0:
            e = torch.rand(rc.shape, generator=g, device=device) < err
            sub = torch.randint(1, 4, rc.shape, generator=g, device=device, dtype=torch.uint8)
            rc = torch.where(e, (rc + sub) & 3, rc)
        flat = rc.reshape(-1).long()
        pad = (-flat.numel()) % 16
        if pad:
            flat = torch.cat([flat, torch.zeros(pad, dtype=torch.int64, device=device)])
        w = (flat.view(-1, 16) << shifts).sum(dim=1)          # < 2^32, disjoint bit fields
        w = torch.where(w >= (1 << 31), w - (1 << 32), w).to(torch.int32)
        w0 = (c0 * read_len) // 16
        words[w0:w0 + w.numel()] = w
        del idx, rc, flat, w
    offsets = torch.arange(n_reads + 1, device=device, dtype=torch.int64) * read_len
    return words, offsets, nwords

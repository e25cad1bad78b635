"""CPU: the minimizer bucket of a k-mer (csrc/sdt_minimizer.cuh through sdt_kmer_final_bucket) -- the function the look-ups of the
multi-GPU owner function call on the device -- against a plain Python restatement: canonical m-mers of the k-mer, the smallest hash,
its second mix, the top 18 bits.  The level-1 scatter files a k-mer under the same value (tests/test_sharded.py pins that half on
the GPU through sdt_kmer_bucket = the top 8 bits).  No reference counterpart: which bucket a k-mer lies in is a layout detail of
this implementation (the reference's is hash_kmer % thrd_num, hashFunction.c:108-122)."""
import numpy as np
import pytest

import __graft_entry__ as ge

M32 = 0xFFFFFFFF


def _mmer_hash(c):
    h = ((c + 0x7F4A7C15) * 0x9E3779B1) & M32
    return h ^ (h >> 15)


def _bucket_hash(h):
    h = ((h ^ 0x5BD1E995) * 0x85EBCA77) & M32
    return h ^ (h >> 13)


def _final_bucket(bases, K):
    m = 11 if K >= 23 else (9 if K >= 17 else 7)
    best = M32
    for p in range(K - m + 1):
        fw = 0
        for b in bases[p:p + m]:
            fw = (fw << 2) | int(b)
        rc = 0
        for b in bases[p:p + m][::-1]:
            rc = (rc << 2) | (int(b) ^ 2)
        best = min(best, _mmer_hash(min(fw, rc)))
    return _bucket_hash(best) >> 14


def _words(bases, K):
    nw = 1 if K <= 31 else (2 if K <= 63 else 4)
    v = 0
    for b in bases:
        v = (v << 2) | int(b)
    return np.array([(v >> (64 * (nw - 1 - i))) & 0xFFFFFFFFFFFFFFFF for i in range(nw)], dtype=np.uint64)


@pytest.mark.parametrize("K", [13, 17, 21, 23, 31, 33, 47, 63, 65, 95, 127])
def test_final_bucket_equals_the_restatement_and_is_strand_independent(K):
    pkg = ge.load_package()
    rng = np.random.default_rng(K)
    for _ in range(40):
        bases = rng.integers(0, 4, size=K)
        if rng.random() < 0.2:
            bases[-(K // 2):] = 3                      # poly-G tails: low words of all ones
        want = _final_bucket(bases, K)
        assert pkg.kmer_final_bucket(_words(bases, K), K) == want
        rcb = (bases[::-1] ^ 2)
        assert pkg.kmer_final_bucket(_words(rcb, K), K) == want, "a k-mer and its reverse complement share their bucket"
        assert pkg.kmer_bucket(_words(bases, K), K) == want >> 10, "the level-1 bucket is the top of the final one"

"""CPU: pin the oracle (oracle/sdt_oracle.c) against fixtures produced by the reference itself.

unit_<variant>.txt come from oracle/_ref/probe<variant> (the reference's kmer.o/hashFunction.o/newhash.o);
cases/* hold what the reference binary wrote.  See tests/golden/make_golden.py."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import golden_util as gu

K4 = ob.Kmer.of


def parse_unit(variant):
    recs = []
    with open(os.path.join(gu.GOLD, f"unit_{variant}.txt")) as fi:
        for line in fi:
            recs.append(line.split())
    return recs


def hx(tok4):
    return tuple(int(t, 16) for t in tok4)


@pytest.mark.parametrize("variant", [31, 63, 127])
def test_unit_vectors(variant):
    L = ob.lib()
    nw = gu.VARIANT_WORDS[variant]
    recs = parse_unit(variant)
    sizes = [r for r in recs if r[0] == "sizeof"][0]
    assert int(sizes[1]) == 8 * nw                      # sizeof(Kmer)
    assert int(sizes[2]) == {1: 24, 2: 32, 4: 48}[nw]   # sizeof(kmer_t): the node record E of SURVEY 8(d)
    nk = 0
    for r in recs:
        if r[0] == "filter":
            assert L.sdto_create_filter(int(r[1])).tup() == hx(r[2:6])
        elif r[0] == "kmer":
            K = int(r[1])
            k = K4(hx(r[2:6]))
            assert r[6] == "rc"
            rc = hx(r[7:11])
            assert L.sdto_reverse_complement(k, K).tup() == rc
            assert r[11] == "hash" and L.sdto_hash_kmer(k, nw) == int(r[12], 16)
            assert r[13] == "smaller" and L.sdto_kmer_smaller(k, K4(rc)) == int(r[14])
            assert r[15] == "first" and L.sdto_first_char(k, K) == int(r[16])
            assert r[17] == "last" and L.sdto_last_char(k) == int(r[18])
            assert r[19] == "next" and L.sdto_next_kmer(k, int(r[20]), K).tup() == hx(r[21:25])
            assert r[25] == "prev" and L.sdto_prev_kmer(k, int(r[26]), K).tup() == hx(r[27:31])
            nk += 1
        elif r[0] == "prime":
            n = int(r[1])
            want = int(r[2])
            got = 3 if n < 3 else L.sdto_next_prime(n)
            assert got == want, (n, got, want)
    assert nk >= 100


def _rng_stream():
    """splitmix64 stream of oracle/ref_probe.c (state continues across the whole probe run, so the table
    part is replayed from the fixture's slot lines instead of the RNG)"""


@pytest.mark.parametrize("variant", [31, 63, 127])
def test_table_layout_replay(variant):
    """put_kmerset / encap_kmerset: rebuild the probe's final table from its own slot dump by inserting
    the distinct keys, then compare sizes, growth sequence and every slot.  Layout is a function of the
    set of keys and their first-occurrence order only; the fixture's slot order is not the
    insertion order, so this checks the weaker, order-free facts; the strict check is in the case tests
    (vertex order) once the graph phases land.  Here: size/max sequence + search finds every key."""
    L = ob.lib()
    nw = gu.VARIANT_WORDS[variant]
    recs = parse_unit(variant)
    init = [r for r in recs if r[0] == "init"][0]
    s = L.sdto_set_new(1024, 0.77)
    st = ob.C.cast(s, ob.C.POINTER(ob.SetStruct)).contents
    assert (st.size, st.max) == (int(init[1]), int(init[2]))
    grows = [(int(r[1]), int(r[2]), int(r[3])) for r in recs if r[0] == "grow"]
    slots = [r for r in recs if r[0] == "slot"]
    final = [r for r in recs if r[0] == "final"][0]
    seen_grow = []
    last = st.size
    for r in slots:
        L.sdto_set_put(s, K4(hx(r[2:6])), 4, 4, nw, None)
        if st.size != last:
            seen_grow.append((st.count, st.size, st.max))
            last = st.size
    # same count -> same (size, max) sequence as the reference run (growth happens at count+1 > max)
    assert [(g[1], g[2]) for g in seen_grow] == [(g[1], g[2]) for g in grows]
    assert (st.size, st.count) == (int(final[2]), int(final[3]))
    slot = ob.C.c_uint64()
    for r in slots:
        assert L.sdto_set_search(s, K4(hx(r[2:6])), nw, ob.C.byref(slot)) == 1
    L.sdto_set_free(s)


def test_base_coding():
    L = ob.lib()
    assert [L.sdto_base2int(ord(c)) for c in "ACTGN"] == [0, 1, 2, 3, 3]     # inc/def.h:39, survey q1
    buf = np.zeros(64, dtype=np.uint8)
    n = L.sdto_encode_line(b"acgtN.x-9R", 10, 100, buf.ctypes.data)
    # lowercase folded; '.' -> A; 'x' is a letter: ('X'&6)>>1 = 0; '-' and '9' dropped; R -> 1
    assert list(buf[:n]) == [0, 1, 3, 2, 3, 0, 0, 1]
    n = L.sdto_encode_line(b"ACGTACGT", 8, 5, buf.ctypes.data)                # truncation to max_rd_len
    assert n == 5


@pytest.mark.parametrize("name", gu.case_names())
def test_case_kmerfreq(name):
    """oracle == reference binary on *.kmerFreq (byte-identical) and on the counters it prints"""
    info = gu.load_case(name)
    variant = info["variant"]
    import importlib
    ge = importlib.import_module("__graft_entry__")
    pkg = ge.load_package()
    K = pkg.clamp_K(info["K"], gu.VARIANT_MAXK[variant])
    codes, offs = gu.case_reads(info)
    o = ob.Oracle(K, nsets=info["p"], nw=gu.VARIANT_WORDS[variant], a=info.get("a", 0))
    o.add_reads(codes, offs)
    assert o.kmers_in_reads() == info["kmer_in_reads"]
    assert o.node_count() == info["nodes_allocated"]
    if info["d"]:
        assert o.delow(info["d"]) == info["kmer_removed"]
    hist, linear = o.mark()
    assert linear == info["linear_nodes"]
    assert ob.kmerfreq_text(hist) == gu.golden_text(info, "kmerFreq")
    assert f" K {K}\n" in gu.golden_text(info, "preGraphBasic")


def run_oracle_pregraph(info, pkg):
    """pass 1 + the three cleaning passes in call_pregraph's order (pregraph.c:63-89)"""
    variant = info["variant"]
    K = pkg.clamp_K(info["K"], gu.VARIANT_MAXK[variant])
    codes, offs = gu.case_reads(info)
    o = ob.Oracle(K, nsets=info["p"], nw=gu.VARIANT_WORDS[variant], a=info.get("a", 0))
    o.add_reads(codes, offs)
    if info["d"]:
        o.delow(info["d"])
    _, linear0 = o.mark()
    counters = {"linear_after": [linear0], "tips_off": []}
    k, ml = o.remove_minor_out(5)
    counters["kmers_off"] = k
    counters["linear_after"].append(ml)
    if not info["d"]:                                  # pregraph.c:73-80: single tips only when -d is off
        t, ml = o.remove_single_tips()
        counters["tips_off"].append(t)
        counters["linear_after"].append(ml)
    t, ml = o.remove_minor_tips()
    counters["tips_off"].append(t)
    counters["linear_after"].append(ml)
    return o, counters


@pytest.mark.parametrize("name", gu.case_names())
def test_case_vertex(name, tmp_path, pkg):
    """minor-out + tip cutting (cutTipPreGraph.c) restated: the counters every pass prints and the final
    *.vertex file (every surviving non-linear node, in the reference's table order) are byte-identical to what
    the reference binary produced at the same -p"""
    info = gu.load_case(name)
    o, c = run_oracle_pregraph(info, pkg)
    assert c["kmers_off"] == info["kmers_off"]
    assert c["tips_off"] == info["tips_off"]
    assert c["linear_after"] == info["linear_after"]
    out = str(tmp_path / "out.vertex")
    assert o.write_vertex(out) == info["vertex_outputed"]
    assert open(out).read() == gu.golden_text(info, "vertex")


@pytest.mark.parametrize("name", gu.case_names())
def test_case_edge_file(name, tmp_path, pkg):
    """kmer2edges restated in full (node2edge.c: startEdgeFromNode, stringBeads, check_iden_kmerList, merge_linearV2;
    output_1edge): after the three cleaning passes the oracle writes the text of *.edge.gz -- every edge in the reference's
    order with its first / last k-mer, length, coverage (including the self-complementary chains whose coverage reads a
    link word already overwritten with the edge id), twin flag and sequence -- byte for byte what the reference binary
    wrote at the same -p, and the counts of its closing line"""
    import gzip
    import re
    info = gu.load_case(name)
    o, _ = run_oracle_pregraph(info, pkg)
    out = str(tmp_path / "out.edge")
    num_ed, emitted, extra = o.write_edges(out)
    with gzip.open(os.path.join(info["dir"], "out.edge.txt.gz"), "rt") as fh:
        want = fh.read()
    assert open(out).read() == want
    m = re.search(r"(\d+) \((\d+)\) edges (\d+) extra nodes", open(os.path.join(info["dir"], "stdout.log")).read())
    assert m and (num_ed, emitted, extra) == tuple(int(x) for x in m.groups())


@pytest.mark.parametrize("name", gu.case_names())
def test_case_prearc(name, tmp_path, pkg):
    """the second read pass restated (prlRead2path.c: parse1read, search1kmerPlus, thread_add1preArc, output_arcs) on top of
    the restated kmer2edges: read -> path of edge ids (linear nodes by their edge id and strand, pairs of vertex nodes through the
    (K+1)-mer patch table -- with the 127mer binary's K = 127 reverse complement) -> arcs; *.preArc byte for byte as the
    reference binary wrote it, list order (most recent first) included"""
    import re
    info = gu.load_case(name)
    o, _ = run_oracle_pregraph(info, pkg)
    o.write_edges(str(tmp_path / "out.edge"))
    codes, offs = gu.case_reads(info)
    out = str(tmp_path / "out.preArc")
    narcs = o.read2edge(codes, offs, out)
    assert open(out).read() == gu.golden_text(info, "preArc")
    m = re.search(r"done mapping reads, \d+ reads deleted, (\d+) arcs created", open(os.path.join(info["dir"], "stdout.log")).read())
    assert m and narcs == int(m.group(1))


def _rc_int(v, K):
    out = 0
    for _ in range(K):
        out = (out << 2) | ((v & 3) ^ 2)
        v >>= 2
    return out


@pytest.mark.parametrize("name", gu.case_names())
def test_case_edges_from_port_walks(name, pkg):
    """kmer2edges' walks restated (sdto_edge_port = startEdgeFromNode + stringBeads + check_iden_kmerList,
    node2edge.c:58-310,563-588) on the graph the three cleaning passes leave: the reference's *.edge.gz lists every edge
    once (its twin is implied, bal_edge = 1) with its first and last oriented k-mer and its length -- walking every port of
    every node that starts edges must find exactly those, each non-palindromic one from both ends"""
    import collections
    import gzip
    info = gu.load_case(name)
    o, _ = run_oracle_pregraph(info, pkg)
    K = o.K

    def rep(frm, to):
        return min((frm, to), (_rc_int(to, K), _rc_int(frm, K)))

    want = collections.Counter()
    with gzip.open(os.path.join(info["dir"], "out.edge.txt.gz"), "rt") as fh:
        for line in fh:
            if not line.startswith(">"):
                continue
            f = line[1:].split(",")                        # length N,<from words>,<to words>,cvg C, bal
            length, bal = int(f[0].split()[1]), int(f[4])
            frm = to = 0
            for w in f[1].split():
                frm = (frm << 64) | int(w, 16)
            for w in f[2].split():
                to = (to << 64) | int(w, 16)
            want[(rep(frm, to), length, bal)] += 2 if bal else 1
    keys = o.export()[0]
    got = collections.Counter()
    for row in keys:
        k = 0
        for x in row:
            k = (k << 64) | int(x)
        for p in range(8):
            w = o.edge_port(row, p)
            if w == -1:
                break
            if w is None:
                continue
            far, aport, length, bal = w
            frm = k if p < 4 else _rc_int(k, K)
            to = far if aport >= 4 else _rc_int(far, K)
            got[(rep(frm, to), length, bal)] += 1
    assert sum(want.values()) > 0
    assert got == want


def test_chop_matches_definition():
    """chopKmer4read restatement vs. the closed form of SURVEY 9.1 (independent of the rolling update)"""
    rng = np.random.default_rng(3)
    L = ob.lib()
    for K in (13, 23, 31, 33, 63, 65, 127):
        nw = ob.key_words_for(K)
        codes = rng.integers(0, 4, size=K + 40, dtype=np.uint8)
        keys, p, q, h = ob.chop_read(codes, K)
        assert len(keys) == len(codes) - K + 1
        for j in range(len(keys)):
            w = 0
            for b in codes[j:j + K]:
                w = (w << 2) | int(b)
            r = 0
            for b in codes[j:j + K][::-1]:
                r = (r << 2) | (int(b) ^ 2)
            has_l, has_r = j > 0, j < len(codes) - K
            if w < r:
                key, pv, nx = w, (codes[j - 1] if has_l else 4), (codes[j + K] if has_r else 4)
            else:
                key, pv, nx = r, ((codes[j + K] ^ 2) if has_r else 4), ((codes[j - 1] ^ 2) if has_l else 4)
            got = 0
            for x in keys[j]:
                got = (got << 64) | int(x)
            assert got == key and p[j] == pv and q[j] == nx
            assert h[j] == L.sdto_hash_kmer(ob.Kmer.of(keys[j]), nw)


# ---- `map` stage: oracle/sdt_oracle_map.c vs the reference's own `map` output ------------
import map_util as mu  # noqa: E402


@pytest.mark.parametrize("name", mu.case_names())
def test_map_oracle_matches_reference_files(tmp_path, name):
    """prlContig2nodes + prlRead2Ctg restated: node / k-mer counters of the contig index and every byte of
    *.readOnContig, *.ctg2Read, *.readInGap (incl. the stale bits of the shared tight-string buffer) and
    *.readInformation (-r) as the reference binary wrote them"""
    info = mu.load_case(name)
    o = mu.build_oracle(info)
    assert o.counts() == (info["nodes_allocated"], info["kmer_in_contigs"])
    codes, offs, lib_of, libs, max_rd_len = mu.case_reads(info)
    counters = o.run(codes, offs, lib_of, [l["avg_ins"] for l in libs], [l["map_len"] for l in libs], max_rd_len, info["p"],
                     tmp_path / "o", trace=bool(info.get("trace")), fill=bool(info.get("fill")))
    assert counters[:3] == [info["reads"], info["reads_mapped"], info["reads_in_gap"]] and counters[3] == 0
    for ext in ["readOnContig", "ctg2Read", "readInGap"] + (["readInformation"] if info.get("trace") else []) + \
            (["shortreadInGap", "PEreadOnContig"] if info.get("fill") else []):
        assert open(str(tmp_path / "o") + "." + ext, "rb").read() == mu.gz_bytes(info, ext), ext

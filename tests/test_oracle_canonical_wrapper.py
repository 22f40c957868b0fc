"""The CanonicalDBG wrapper over PRIMARY DBGSuccinct graphs restated in the oracle (graph/representation/canonical_dbg.cpp:
node space base + offset, map_to_nodes_sequentially :55-146, call_outgoing_kmers / call_incoming_kmers through the reverse
complement's (k-1)-mer range :156-330,574-684, reverse_complement :515-560) against the wrapper's own KATs
(tests/graph/test_canonical_dbg.cpp: InsertSequence :73-85, ReverseComplement :87-101, Traversals1 :103-158, Traversals2
:160-195, the dummy k-mer traversals :1239-1640).  SURVEY 8(f) rank 1; the aligner on this view: below, tests/test_oracle_primary_goldens.py and test_emu_primary.py.  The reference's test helper builds
PRIMARY graphs from primary contigs; here the inputs are chosen so that the sequences themselves hold one k-mer of every
pair (the assertions do not depend on which one)."""
import ctypes as C

import pytest

import orc

PRIMARY = 2


def _L():
    L = orc.L()
    if not getattr(L, "_canon_ready", False):
        L.orc_canonical_adjacent.restype = C.c_uint32
        L.orc_canonical_adjacent.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64), C.c_char_p]
        L.orc_canonical_reverse_complement.restype = C.c_uint64
        L.orc_canonical_reverse_complement.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_canonical_node_sequence.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p]
        L.orc_canonical_map.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint64)]
        L.orc_canonical_degrees.restype = C.c_int
        L.orc_canonical_degrees.argtypes = [C.c_void_p, C.c_uint64]
        L._canon_ready = True
    return L


def cmap(g, seq):
    n = len(seq) - g.k + 1
    out = (C.c_uint64 * max(1, n))()
    _L().orc_canonical_map(g.h, seq.encode(), len(seq), out)
    return [out[i] for i in range(max(0, n))]


def adjacent(g, v, direction):
    nodes = (C.c_uint64 * 8)()
    chars = C.create_string_buffer(8)
    n = _L().orc_canonical_adjacent(g.h, v, direction, nodes, chars)
    return [(nodes[i], chars.raw[i:i + 1].decode()) for i in range(n)]


def traverse(g, v, c):           # CanonicalDBG::traverse == the child reached over c
    hits = [n for n, ch in adjacent(g, v, 0) if ch == c]
    assert len(hits) <= 1
    return hits[0] if hits else 0


def traverse_back(g, v, c):
    hits = [n for n, ch in adjacent(g, v, 1) if ch == c]
    assert len(hits) <= 1
    return hits[0] if hits else 0


def node_seq(g, v):
    buf = C.create_string_buffer(g.k)
    _L().orc_canonical_node_sequence(g.h, v, buf)
    return buf.raw[:g.k].decode()


def rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def find(g, seq):
    nodes = cmap(g, seq)
    return bool(nodes) and all(nodes)


def test_insert_sequence_and_reverse_complement():
    # :73-101
    g = orc.Graph.build(21, ["AAAAAAAAAAAAAAAAAAAAAAAAAAAAA", "CATGTACTAGCTGATCGTAGCTAGCTAGC"], 0, True)
    assert find(g, "AAAAAAAAAAAAAAAAAAAAAAAAAAAAA") and find(g, "TTTTTTTTTTTTTTTTTTTTTTTTTTTTT")
    assert find(g, "CATGTACTAGCTGATCGTAGCTAGCTAGC") and find(g, "GCTAGCTAGCTACGATCAGCTAGTACATG")
    assert not find(g, "CATGTTTTTTTAATATATATATTTTTAGC") and not find(g, "GCTAAAAATATATATATTAAAAAAACATG")
    # every k-mer and its mirror: ids differ by the offset, spellings are reverse complements
    fwd = cmap(g, "CATGTACTAGCTGATCGTAGCTAGCTAGC")
    rev = cmap(g, "GCTAGCTAGCTACGATCAGCTAGTACATG")
    for a, b in zip(fwd, reversed(rev)):
        assert _L().orc_canonical_reverse_complement(g.h, a) == b and _L().orc_canonical_reverse_complement(g.h, b) == a
        assert node_seq(g, b) == rc(node_seq(g, a))


@pytest.mark.parametrize("k", range(2, 10))
def test_traversals1(k):
    # :103-158
    g = orc.Graph.build(k, ["A" * 100 + "C" * 100], 0, True)
    it, jt = cmap(g, "A" * k)[-1], cmap(g, "T" * k)[-1]
    assert it and jt and _L().orc_canonical_reverse_complement(g.h, it) == jt
    it2 = cmap(g, "A" * (k - 1) + "C")[-1]
    jt2 = cmap(g, "G" + "T" * (k - 1))[-1]
    assert it2 and jt2
    assert traverse(g, it, "A") == it and traverse(g, it, "C") == it2 and traverse_back(g, it2, "A") == it
    assert traverse(g, jt, "T") == jt and traverse_back(g, jt, "G") == jt2 and traverse(g, jt2, "T") == jt
    assert traverse(g, it, "G") == 0 and traverse_back(g, it2, "G") == 0
    it = cmap(g, "G" * k)[-1]
    assert it and cmap(g, "C" * k)[-1] == _L().orc_canonical_reverse_complement(g.h, it)
    it2 = cmap(g, "G" * (k - 1) + "T")[-1]
    assert it2
    assert traverse(g, it, "A") == 0 and traverse(g, it, "G") == it and traverse(g, it, "T") == it2
    assert traverse_back(g, it2, "G") == it


@pytest.mark.parametrize("k", range(2, 11))
def test_traversals2(k):
    # :160-195 (the second input sequence of the reference test is the reverse complement of the first: one strand suffices)
    g = orc.Graph.build(k, ["A" * 100 + "C" * 100], 0, True)
    it = cmap(g, "A" * k)[-1]
    assert it and traverse(g, it, "A") == it
    nxt = traverse(g, it, "C")
    assert nxt and nxt != it and traverse_back(g, nxt, "A") == it
    assert traverse(g, it, "G") == 0 and traverse_back(g, it, "G") == 0
    it = cmap(g, "G" * k)[-1]
    assert it and traverse(g, it, "G") == it
    assert traverse(g, it, "T") == cmap(g, "G" * (k - 1) + "T")[-1]
    assert traverse_back(g, traverse(g, it, "T"), "G") == it


# ---- dummy k-mers seen through the wrapper (test_canonical_dbg.cpp:1239-1640; unmasked DBGSuccinct(31, PRIMARY)) ----
DUMMY_GRAPHS = {
    "sink_source": ["CTTCCTTCCTTCTTTCCTTCCTTCCTTCCTC"],
    "dummy": ["CTTCCTTCCTTCTTTCCTTCCTTCCTTCCTC", "AAGGAAGGAAGGAAGGAAAGAAGGAAGGAAG"],
    "to_dummy_fwd": ["TGTGCGGCGGGAATATGTACGAAGCGCAGGA", "CCTGCGCTTCGTACATATTCCCGCCGCACAG"],
    "to_dummy_bwd": ["TGTGCGGCGGGAATATGTACGAAGCGCAGGA", "TTCCTGCGCTTCGTACATATTCCCGCCGCAC"],
    "no_dual": ["TCCTGCGCTTCGTACATATTCCCGCCGCACT", "TGTGCGGCGGGAATATGTACGAAGCGCAGGA"],
    "no_dup_sink": ["TCCTGCGCTTCGTACATATTCCCGCCGCACT", "AGTGCGGCGGGAATATGTACGAAGCGCAGGC"],
}


def dummy_graph(name):
    return orc.Graph.build(31, DUMMY_GRAPHS[name], PRIMARY, False)


def base_node_spelled(g, spelling):
    """the base graph's node (edge) with this spelling, sentinels included (the reference looks it up with index_range + bwd)"""
    hits = [v for v in range(1, g.n_edges + 1) if g.node_sequence(v) == spelling]
    assert len(hits) == 1, (spelling, hits)
    return hits[0]


def base_adjacent(g, v, incoming):
    return g.outgoing(v, rc=False) if not incoming else [(n, rc(c) if c != "$" else c) for n, c in g.outgoing(v, rc=True)]


def test_traversal_dummy_sink_and_source():
    # :1239-1344: the only neighbours of the first k-mer of a lone sequence are dummy k-mers, in the base graph and through the wrapper
    g = dummy_graph("sink_source")
    node = cmap(g, "CTTCCTTCCTTCTTTCCTTCCTTCCTTCCTC")[0]
    assert node
    out = adjacent(g, node, 0)
    assert [c for _, c in out] == ["$"] and node_seq(g, out[0][0])[-1] == "$"
    inc = adjacent(g, node, 1)
    assert [c for _, c in inc] == ["$"] and node_seq(g, inc[0][0])[0] == "$"


def test_traversal_dummy():
    # :1346-1395: a dummy source k-mer $X has its own child (C) and one more through the reverse complement (T)
    g = dummy_graph("dummy")
    v = base_node_spelled(g, "$" + "CTTCCTTCCTTCTTTCCTTCCTTCCTTCCT")
    assert node_seq(g, v) == "$CTTCCTTCCTTCTTTCCTTCCTTCCTTCCT"
    assert [c for _, c in g.outgoing(v)] == ["C"]
    assert sorted(c for _, c in adjacent(g, v, 0)) == ["C", "T"]


def test_traverse_no_dummy_to_dummy_forward_and_backward():
    # :1397-1498: no step from a dummy source k-mer to the reverse complement of a dummy sink k-mer, in either direction
    g = dummy_graph("to_dummy_fwd")
    v = base_node_spelled(g, "$" + "TGTGCGGCGGGAATATGTACGAAGCGCAGG")
    assert [c for _, c in g.outgoing(v)] == ["A"]
    out = adjacent(g, v, 0)
    assert out and all(c != "$" and node_seq(g, n)[-1] != "$" for n, c in out)
    g = dummy_graph("to_dummy_bwd")
    sink = base_node_spelled(g, "GTGCGGCGGGAATATGTACGAAGCGCAGGA" + "$")
    node = _L().orc_canonical_reverse_complement(g.h, sink)
    out = adjacent(g, node, 0)
    assert out and all(c == "A" and node_seq(g, n)[-1] == "A" for n, c in out)


def test_traverse_no_dual_dummy_and_no_duplicate_sink():
    # :1500-1640: the reverse complements of dummy k-mers are not neighbours: exactly one sentinel callback per side
    g = dummy_graph("no_dual")
    v = base_node_spelled(g, "TGTGCGGCGGGAATATGTACGAAGCGCAGGA")
    out = adjacent(g, v, 0)
    assert sum(c == "$" for _, c in out) == 1 and sum(node_seq(g, n)[-1] == "$" for n, _ in out) == 1
    v = base_node_spelled(g, "TCCTGCGCTTCGTACATATTCCCGCCGCACT")
    inc = adjacent(g, v, 1)
    assert sum(c == "$" for _, c in inc) == 1 and sum(node_seq(g, n)[0] == "$" for n, _ in inc) == 1
    g = dummy_graph("no_dup_sink")
    start = cmap(g, "TCCTGCGCTTCGTACATATTCCCGCCGCACT")[0]
    out = adjacent(g, start, 0)
    assert all(c == "$" for _, c in out) and len({n for n, _ in out}) == len(out)
    start = cmap(g, "AGTGCGGCGGGAATATGTACGAAGCGCAGGC")[0]
    inc = adjacent(g, start, 1)
    assert all(c == "$" for _, c in inc) and len({n for n, _ in inc}) == len(inc)


# ---- the aligner on the wrapper (PRIMARY graph + CanonicalDBG): the reference's PRIMARY KATs ----
from metagraph_amd import capi  # noqa: E402


def _cfg(**kw):
    c = capi.config_default()
    capi.set_dna_matrix(c, 2, -1, -2)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def test_align_suffix_seed_snp_canonical_primary_half():
    # test_aligner.cpp:1483-1537, the DeBruijnGraph::PRIMARY iteration: DBGSuccinct(k, PRIMARY) + add_sequence(reference_rc),
    # wrapped into CanonicalDBG; the query matches the reverse complement of what the graph stores
    k = 18
    reference, query = "AAAAACTTTCGAGGCCAA", "GGGGGCTTTCGAGGCCAA"
    reference_rc = "TTGGCCTCGAAAGTTTTT"
    g = orc.Graph.build(k, [reference_rc], PRIMARY, False)
    cfg = _cfg(max_num_seeds_per_locus=capi.UINT64_MAX, min_cell_score=-2147483648 + 100, min_path_score=-2147483648 + 100,
               min_seed_length=13)
    run = orc.AlignRun(g, cfg, [query])
    assert run.error == "", run.error
    paths = run.results()[0]
    assert len(paths) == 1
    p = paths[0]
    assert len(p["nodes"]) == 1
    if p["sequence"] == reference[5:]:
        assert p["cigar"] == "5S13=" and p["clipping"] == 5 and p["end_clipping"] == 0 and p["score"] == 26
    else:
        assert p["sequence"] == reference_rc[:13]
        assert p["cigar"] == "13=5S" and p["clipping"] == 0 and p["end_clipping"] == 5 and p["score"] == 26
    assert p["offset"] == 5 and p["num_matches"] == 13


@pytest.mark.parametrize("max_seed_length", [0, 31 + 100])
def test_align_suffix_seed_no_full_seeds(max_seed_length):
    # test_aligner.cpp:1775-1800: one alignment, valid on the wrapped graph (is_valid runs inside AlignRun), with and
    # without full-k-mer seeding
    k = 31
    reference = "CTGCTGCGCCATCGCAACCCACGGTTGCTTTTTGAGTCGCTGCTCACGTTAGCCATCACACTGACGTTAAGCTGGCTTTCGATGCTGTATC"
    query = "CTTACTGCTGCGCTCTTCGCAAACCCCACGGTTTCTTGTTTTGAGCTCGCCTGCTCACGATACCCATACACACTGACGTTCAAGCTGGCTTTCGATGTTGTATC"
    g = orc.Graph.build(k, [reference], PRIMARY, False)
    cfg = _cfg(max_num_seeds_per_locus=capi.UINT64_MAX, min_cell_score=-2147483648 + 100, min_path_score=-2147483648 + 100,
               min_seed_length=13, max_seed_length=max_seed_length)
    run = orc.AlignRun(g, cfg, [query])
    assert run.error == "", run.error
    assert len(run.results()[0]) == 1

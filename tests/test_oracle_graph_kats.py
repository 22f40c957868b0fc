import os
"""Known-answer tests of the reference for DBGSuccinct::call_nodes_with_suffix_matching_longest_prefix
(M/tests/graph/succinct/test_dbg_succinct.cpp:162-527), transcribed as data and run against the oracle.
This is the primitive behind the seeder's sub-k seeds (SURVEY 8a rows a12/a13)."""
import pytest

import orc

# (name, k, sequences added one by one, mask_dummy_kmers?, string passed, min_match_length, expected node strings,
#  expected match length); source lines in test_dbg_succinct.cpp
SUFFIX_KATS = [
    ("CallNodesWithSuffix", 4, ["GGCCCAGGGGTC"], True, "GG", 1, ["AGGG", "CAGG", "GGGG"], 2),                  # :162-198
    ("CallNodesWithSuffixMinLength", 4, ["GGCCCAGGGGTC"], True, "CAGC", 4, [], 4),                            # :200-225
    ("CallNodesWithSuffixK", 4, ["GGCCCAGGGGTC"], True, "GGCC", 1, ["GGCC"], 4),                              # :227-259
    ("CallNodesWithSuffixK_v2", 4, ["AGCCC", "CGCC", "GGCC", "TGCC"], False, "AGCC", 4, 1, 4),                # :261-285 (count only)
    ("CallNodesWithSuffixKEarlyCutoff", 4, ["GGCCCAGGGGTC"], True, "GG", 1, ["AGGG", "CAGG", "GGGG"], 2),     # :287-323 (query[:2])
    ("CallNodesWithSuffixEarlyCutoffKMinusOne", 4, ["GGCCCAGGGGTC"], True, "GGG", 1, ["AGGG", "GGGG"], 3),    # :325-359 (query[:3])
    ("CallNodesWithSuffixKMinusOne", 4, ["GGCCCAGGGGTC"], True, "GCC", 1, ["GGCC"], 3),                       # :361-393
    ("CallNodesWithSuffixKMinusOneBeginning", 4, ["GGCCCAGGGGTC"], True, "GGC", 1, [], 3),                    # :395-419
    ("CallNodesWithSuffixMinusTwoBeginning", 4, ["TGCCCAGGGGTC"], True, "TG", 1, [], 2),                      # :421-445
    ("CallNodesWithSuffixMultipleOut", 3, ["GGGGGGATGTAG", "GGGGGGATGCCTAATTAA"], True, "TGC", 1, ["TGC"], 3),  # :447-479
    ("CallNodesWithSuffixMultipleInOut", 4, ["AAAAAAAAATGC", "GGGGGGGGATGG", "GGGGGGGGTTGC", "AAAAAAAATTGG"], True,
     "TGA", 1, ["AATG", "GATG", "GTTG", "ATTG"], 2),                                                           # :481-527
]


@pytest.mark.parametrize("case", SUFFIX_KATS, ids=lambda c: c[0])
def test_call_nodes_with_suffix_matching_longest_prefix(case):
    name, k, seqs, mask, query, min_len, want, want_len = case
    g = orc.Graph.build(k, seqs, 0, mask)
    nodes, match_len = g.suffix_match(query, min_len)
    if isinstance(want, int):
        assert len(nodes) == want
    else:
        assert sorted(g.node_sequence(v) for v in nodes) == sorted(want)
        for v in nodes:
            s = g.node_sequence(v)
            assert query[:want_len] == s[len(s) - want_len:] or name == "CallNodesWithSuffixMultipleOut"
    if nodes:
        assert match_len == want_len


def test_call_outgoing_kmers_source():
    """DBGSuccinct.call_outgoing_kmers_source (test_dbg_succinct.cpp:138-160): the children of the source dummy
    edge 1 in an unmasked graph, with their node ids."""
    g = orc.Graph.build(4, ["AATGG", "CCGAA"], 0, False)
    assert sorted(g.outgoing(1)) == [(2, "A"), (3, "C")]


@pytest.mark.parametrize("k", range(2, 10))
def test_source_dummy_is_node_one_until_masked(k):
    """get_degree_with_source_dummy (test_dbg_succinct.cpp:13-56), the parts on the alignment path: edge 1 spells
    '$' * k; after mask_dummy_kmers it is no longer a node, and the real k-mers keep their out-neighbours."""
    seq = "A" * k + "C" * (k - 1) + "G" * (k - 1) + "T" * k
    g = orc.Graph.build(k, [seq], 0, False)
    assert g.node_sequence(1) == "$" * k
    # (the reference test builds with add_sequence, which keeps the redundant source-dummy path $$$A.. into AAAA;
    # the fixture builder follows construct_boss_chunk, boss_chunk_construct.cpp:57-171, which removes it — so the
    # in-degree assertions of that test do not transfer, the out-neighbours do)
    gm = orc.Graph.build(k, [seq], 0, True)
    W, last, F, valid = gm.export()
    assert valid is not None and not valid[1]
    for graph in (g, gm):
        nodes, ml = graph.suffix_match("A" * k, k)
        assert ml == k and len(nodes) == 1
        assert sorted(c for _, c in graph.outgoing(nodes[0])) == ["A", "C"]      # outdegree(AAAA) == 2


@pytest.mark.parametrize("kb", range(1, 10))
def test_boss_map_to_edges(kb):
    """BOSS.map_to_edges (M/tests/graph/succinct/test_boss.cpp:2317-2345).  The reference builds this graph with
    add_sequence, which keeps the kb redundant source-dummy edges $..$A.. in front of A^(kb+1); the fixture builder
    follows construct_boss_chunk and removes them (A^(kb+1) has a real predecessor), so every edge index is lower by
    exactly kb.  Positions, npos entries and the restart/walk pattern are the reference's."""
    from metagraph_amd import capi
    k = kb + 1
    g = orc.Graph.build(k, ["A" * 100 + "C" * 100], 0, False)
    seq = "T" * 2 + "A" * (kb + 3) + "C" * (2 * kb)
    expected = [0, 0, kb + 2, kb + 2, kb + 2] + [kb + 2 + i for i in range(1, kb + 1)] + [kb + 2 + kb + 1] * kb
    got = orc.AlignRun(g, capi.config_cli(k), [seq]).mapping()[0][0]
    assert got == [e - kb if e else 0 for e in expected]


# ---- node degrees (M/tests/graph/all/test_dbg_node_degree.cpp, DBGSuccinct instantiation: build_graph masks the
# dummy k-mers, test_dbg_helpers.cpp:379).  outdegree / indegree are counted through the traversal functions the
# aligner uses (call_outgoing_kmers, RCDBG's call_outgoing_kmers == incoming), and has_multiple_outgoing /
# has_single_incoming — the UniMEM terminus test of the seeder — must agree with them (check_degree_functions :48-80).
def _node(g, kmer):
    nodes, ml = g.suffix_match(kmer, len(kmer))
    assert ml == len(kmer) and len(nodes) == 1, kmer
    return nodes[0]


def _degrees(g, v):
    return len(g.outgoing(v)), len(g.outgoing(v, rc=True))


def _check_degree_functions(g):
    W, last, F, valid = g.export()
    for v in range(1, len(W)):
        if valid is not None and not valid[v]:
            continue
        outd, ind = _degrees(g, v)
        assert bool(orc.L().orc_graph_has_multiple_outgoing(g.h, v)) == (outd > 1), v
        assert bool(orc.L().orc_graph_has_single_incoming(g.h, v)) == (ind == 1), v


@pytest.mark.parametrize("k", range(2, 10))
def test_node_degree_kats(k):
    g = orc.Graph.build(k, ["A" * (k - 1) + "C"], 0, True)                      # get_outdegree/indegree_single_node :18-24,105-114
    assert g.num_nodes == 1 and _degrees(g, _node(g, "A" * (k - 1) + "C")) == (0, 0)
    _check_degree_functions(g)
    g = orc.Graph.build(k, ["A" * (k - 1) + c for c in "ACGT"], 0, True)        # get_maximum_outdegree :26-46
    assert g.num_nodes == 4
    for c in "ACGT":
        assert _degrees(g, _node(g, "A" * (k - 1) + c))[0] == (4 if c == "A" else 0)
    g = orc.Graph.build(k, [c + "A" * (k - 1) for c in "ACGT"], 0, True)        # get_maximum_indegree :116-138
    assert g.num_nodes == 4
    for c in "ACGT":
        assert _degrees(g, _node(g, c + "A" * (k - 1)))[1] == (4 if c == "A" else 0)
    _check_degree_functions(g)
    g = orc.Graph.build(k, ["A" * k + "C" * (k - 1) + "G" * (k - 1) + "T" * k], 0, True)        # get_degree1 :179-201
    assert _degrees(g, _node(g, "A" * k)) == (2, 1) and _degrees(g, _node(g, "T" * k)) == (1, 2)
    _check_degree_functions(g)
    g = orc.Graph.build(k, ["A" * k + "C" * (k - 1) + "G" * (k - 1) + "T" * (k - 1)], 0, True)  # get_degree2 :203-225
    assert _degrees(g, _node(g, "A" * k)) == (2, 1) and _degrees(g, _node(g, "G" + "T" * (k - 1))) == (0, 1)
    _check_degree_functions(g)


def test_node_degree_indegree1():
    g = orc.Graph.build(3, ["ACTAAGCCC", "AAAGC", "TAAGCA"], 0, True)           # indegree1 :164-177
    assert g.num_nodes == 9
    assert _degrees(g, _node(g, "CCC")) == (1, 2)
    assert _degrees(g, _node(g, "AAA")) == (2, 2)
    _check_degree_functions(g)


# ---- call_outgoing_kmers / call_incoming_kmers (M/tests/graph/all/test_dbg_traverse.cpp:106-298): the characters
# reported for each node, and that the reported neighbour spells node[1:] + c (resp. c + node[:-1]).  The incoming side
# is read through the RC view the aligner uses (rc_dbg.hpp:88-99 complements the first character).
COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def _out_chars(g, v):
    seq = g.node_sequence(v)
    res = g.outgoing(v)
    for n, c in res:
        assert g.node_sequence(n) == seq[1:] + c
    return sorted(c for _, c in res)


def _in_chars(g, v):
    seq = g.node_sequence(v)
    res = [(n, COMP[c]) for n, c in g.outgoing(v, rc=True)]
    for n, c in res:
        assert g.node_sequence(n) == c + seq[:-1]
    return sorted(c for _, c in res)


@pytest.mark.parametrize("k", range(3, 11))
def test_call_outgoing_edges(k):
    g = orc.Graph.build(k, ["A" * 100 + "C" * 100 + "G" * (k - 1)], 0, True)
    assert _out_chars(g, _node(g, "A" * k)) == ["A", "C"]
    assert _out_chars(g, _node(g, "A" * (k - 1) + "C")) == ["C"]
    assert _out_chars(g, _node(g, "C" * k)) == ["C", "G"]
    assert _out_chars(g, _node(g, "C" * (k - 1) + "G")) == ["G"]
    assert _out_chars(g, _node(g, "C" + "G" * (k - 1))) == []
    assert g.suffix_match("G" * k, k)[0] == []                       # GGG does not exist


@pytest.mark.parametrize("k", range(3, 11))
def test_call_incoming_edges(k):
    g = orc.Graph.build(k, ["A" * (k - 1) + "C" * 100 + "G" * (k - 1)], 0, True)
    assert g.suffix_match("G" * k, k)[0] == []
    assert _in_chars(g, _node(g, "C" + "G" * (k - 1))) == ["C"]
    assert _in_chars(g, _node(g, "C" * (k - 1) + "G")) == ["A", "C"]
    assert _in_chars(g, _node(g, "C" * k)) == ["A", "C"]
    assert _in_chars(g, _node(g, "A" + "C" * (k - 1))) == ["A"]
    assert _in_chars(g, _node(g, "A" * (k - 1) + "C")) == []
    assert g.suffix_match("A" * k, k)[0] == []


@pytest.mark.parametrize("k", range(2, 11))
def test_map_to_nodes_whole_sequence_equals_per_kmer(k):
    """DeBruijnGraphTest.map_to_nodes (M/tests/graph/all/test_dbg_search.cpp:101-133): mapping a sequence with misses at
    its start equals mapping every k-mer on its own (the incremental fwd + pick_edge walk and the re-index after a miss
    agree with index() from scratch)."""
    from metagraph_amd import capi
    g = orc.Graph.build(k, ["A" * 100 + "C" * 100], 0, True)
    seq = "T" * 2 + "A" * (k + 2) + "C" * (2 * (k - 1))
    cfg = capi.config_cli(k)
    whole = orc.AlignRun(g, cfg, [seq]).mapping()[0][0]
    kmers = [seq[i:i + k] for i in range(len(seq) - k + 1)]
    single = [m[0][0] for m in orc.AlignRun(g, cfg, kmers).mapping()]
    assert whole == single
    assert whole[0] == 0 and whole[1] == 0 and all(v != 0 for v in whole[2:])


def test_build_stats_of_transcripts_1000_k20():
    """integration_tests/test_build.py:30-48: `metagraph build --mask-dummy -k 20` on tests/data/transcripts_1000.fa reports
    591997 nodes (k); the fixture builder's masked graph must hold exactly as many real k-mers."""
    from test_oracle_kats import read_fasta, HERE
    g = orc.Graph.build(20, read_fasta(os.path.join(HERE, "golden", "transcripts_1000.fa")), 0, True)
    assert g.num_nodes == 591997


def test_build_stats_of_canonical_genome_mt_k11():
    """integration_tests/test_align.py:209-219: `metagraph build --mode canonical --mask-dummy -k 11` on genome.MT.fa reports
    32782 nodes (k) (basic mode: 16438, already pinned).  The fixture builder's CANONICAL mode adds the reverse complement
    of every sequence (boss_chunk_construct.cpp:357-359): groundwork for canonical / primary graphs."""
    from test_oracle_kats import read_fasta, HERE
    g = orc.Graph.build(11, read_fasta(os.path.join(HERE, "golden", "genome.MT.fa")), 1, True)
    assert g.num_nodes == 32782

"""Post-alignment chaining (chain_alignments, aligner_chainer.cpp:555-720; config.post_chain_alignments): the reference's own
tests (tests/graph/test_aligner_chain.cpp:36-269) restated on the oracle — their assertions are the number of paths, the
spelling of the chain ('$' where two alignments are joined across a gap) and Alignment::is_valid — and the same queries
through the kernels' host model with the library's host-side chaining (libmgx does the chaining on the host: it works on the
handful of alignments a query has left)."""
import pytest

import orc
from metagraph_amd import capi

# (name, k, references, query, (match, mismatch_transition, mismatch_transversion), gaps or None, expected spelling, is a chain)
CASES = [
    ("align_chain_swap", 5, ["ATGATATGATGACCCCGG"], "TGACCCCGGATGATATGA", (2, -1, -2), None, "TGACCCCGGATGATATGA", True),
    ("align_chain_overlap_2", 5, ["TGAGGATCAG", "CAGCTAGCTAGCTAGC"], "TGAGGATCAGCTAGCTAGCTAGC", (2, -1, -2), None,
     "TGAGGATCAGCTAGCTAGCTAGC", True),
    ("align_chain_overlap_3_prefer_mismatch_over_gap", 5, ["TGAGGATCAG", "CAGCTAGCT", "GCTTGCTAGC"], "TGAGGATCAGCTAGCTTGCTAGC",
     (2, -3, -3), None, "TGAGGATCAGCTAGCTAGCTAGC", True),
    ("align_chain_insert_no_chain_if_full_coverage", 10, ["TGAGGATCAGTTCTAGCTTGCTAGC"], "TGAGGATCAGCTAGCTTGCTAGC", (2, -1, -2), None,
     "TGAGGATCAGTTCTAGCTTGCTAGC", False),
    ("align_chain_insert1", 10, ["TGAGGATCAGTTCTAGCTTG", "CTAGCTTGCTAGCGCTAGCTAGATC"], "TGAGGATCAGCTAGCTTGCTAGCGCTAGCTAGATC", (2, -1, -2),
     None, "TGAGGATCAGTTCTAGCTTGCTAGCGCTAGCTAGATC", True),
    ("align_chain_insert_mismatch", 10, ["TGAGGATCAGTTCTAGCTTG", "CTAGCTTGCTAGCGCTAGCTAGATC"], "TGAGGATCAGCTTGCTTGCTAGCGCTAGCTAGATC",
     (2, -1, -2), None, "TGAGGATCAGTTCTAGCTTGCTAGCGCTAGCTAGATC", True),
    ("align_chain_insert_in_overlap", 10, ["TGAGGATCAGTTCTAGCTTG", "CTAGCTTGCTAGCGCTAGCTAGATC"], "TGAGGATCAGCTAAGCTTGCTAGCGCTAGCTAGATC",
     (2, -1, -2), None, "TGAGGATCAGTTCTAGCTTGCTAGCGCTAGCTAGATC", True),
    ("align_chain_large_overlap", 10, ["TGAGGATCAGTTCTAGCTTG", "ATCAGTTCTAGCTTGCTAGCGCTAGCTAGATC"],
     "TGAGGATCAGTAATCTAGCTTGCTAGCGCTAGCTAGATC", (2, -1, -2), None, "TGAGGATCAGTTCTAGCTTGCTAGCGCTAGCTAGATC", False),
    ("align_chain_overlap_with_insert", 10, ["TGAGGATCAGTTCTAGCTTG", "CTAGCTTGCTAGCGCTAGCTAGATC"],
     "TGAGGATCAGTTCTAAGCTTGCTAGCGCTAGCTAGATC", (1, -1, -1), (-1, -1), "TGAGGATCAGTTCTAGCTTGCTAGCGCTAGCTAGATC", True),
    ("align_chain_delete_in_overlap", 10, ["TGAGGATCAGTTCTAGCTTG", "CTAGCTTGCTAGCGCTAGCTAGATC"], "TGAGGATCAGTTCTACTTGCTAGCGCTAGCTAGATC",
     (2, -1, -2), None, "TGAGGATCAGTTCTAGCTTGCTAGCGCTAGCTAGATC", True),
    ("align_chain_disjoint", 10, ["CCCCCCCCTGAGGATCAG", "TTCACTAGCTAGCCCCCCCCC"], "CCCCCCCCTGAGGATCAGTTCACTAGCTAGCCCCCCCCC", (2, -1, -2), None,
     "CCCCCCCCTGAGGATCAG$TTCACTAGCTAGCCCCCCCCC", True),
    ("align_chain_gap", 10, ["AAAAACCCCCTGAGGATCAG", "ACTAGCTAGCCCCCCAAAAA"], "AAAAACCCCCTGAGGATCAGTTCACTAGCTAGCCCCCCAAAAA", (1, -1, -1),
     (-1, -1), "AAAAACCCCCTGAGGATCAG$ACTAGCTAGCCCCCCAAAAA", True),
]


def chain_config(k, scores, gaps):
    # DBGAlignerConfig defaults (aligner_config.hpp:18-61) + the test's scoring + post_chain_alignments
    c = capi.config_default()
    capi.set_dna_matrix(c, *scores)
    if gaps:
        c.gap_opening_penalty, c.gap_extension_penalty = gaps
    c.post_chain_alignments = 1
    return c


def chain_graph(k, refs):
    # std::make_shared<DBGSuccinct>(k) + add_sequence: dummy k-mers are part of the graph (no mask)
    return orc.Graph.build(k, refs, 0, False)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_reference_chain_tests_on_the_oracle(case):
    name, k, refs, query, scores, gaps, want, is_chain = case
    g = chain_graph(k, refs)
    cfg = chain_config(k, scores, gaps)
    run = orc.AlignRun(g, cfg, [query])           # (validate=True: Alignment::is_valid on every path, as check_chain does)
    assert run.error == "", run.error
    (paths,) = run.results()
    assert len(paths) == 1, paths
    assert paths[0]["sequence"] == want, paths[0]
    # check_chain: a chain cannot be dumped to JSON (to_json throws on a path with dummy nodes), a plain alignment can
    assert (0 in paths[0]["nodes"]) == is_chain, paths[0]


def test_post_chaining_off_is_a_pass_through():
    name, k, refs, query, scores, gaps, want, is_chain = CASES[4]
    g = chain_graph(k, refs)
    cfg = chain_config(k, scores, gaps)
    cfg.post_chain_alignments = 0
    (paths,) = orc.AlignRun(g, cfg, [query]).results()
    assert len(paths) == 1 and 0 not in paths[0]["nodes"] and paths[0]["sequence"] != want


def product_chain(g, k, cfg, queries, limits=None):
    """the product's two halves on the CPU: the kernels' host model with the keep-every-alignment aggregator
    (DevConfig::post_chain), then libmgx's host-side chaining (mgx_chain_alignments: plain host code)"""
    import ctypes as C
    import emu_drv
    eg = emu_drv.EmuGraph(g)
    e = emu_drv.EmuRun(eg, cfg, queries, limits=limits)
    assert e.error == "", e.error
    plain, status = e.results()
    assert all(s in (0, capi.MGX_ERR_CAPACITY) for s in status), status
    v = capi.Results()
    emu_drv.L().emu_results(e.r, C.byref(v))
    chained = capi.chain_alignments(cfg, k, v, queries)
    # a query the device could not keep every alignment of (more than 4 x MGX_MAX_ALTERNATIVE_PATHS - 3: mgx.h) has a capacity
    # status and no alignments — never a shortened list
    for q, st in enumerate(status):
        if st:
            assert plain[q] == [] and chained[q] == []
            chained[q] = None
    return plain, chained


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_reference_chain_tests_through_the_kernels_and_the_host_chaining(case):
    name, k, refs, query, scores, gaps, want, is_chain = case
    g = chain_graph(k, refs)
    cfg = chain_config(k, scores, gaps)
    (want_paths,) = orc.AlignRun(g, cfg, [query]).results()
    plain, (got_paths,) = product_chain(g, k, cfg, [query])
    assert got_paths == want_paths, (plain, got_paths, want_paths)
    assert got_paths[0]["sequence"] == want


def test_chaining_random_worlds_against_the_oracle():
    """reads stitched from two places of a genome (and their reverse complements), with errors: whatever the aligner leaves —
    one alignment, several, chains with overlaps, gaps, dummy nodes — the product's list equals the oracle's"""
    import random
    from test_emu_vs_oracle import rand_seq, mutate, rc
    n_chained = n_capacity = 0
    for seed in range(12):
        rng = random.Random(500 + seed)
        k = rng.choice([10, 12, 15])
        genome = rand_seq(rng, 1500)
        g = orc.Graph.build(k, [genome], 0, False)
        cfg = chain_config(k, (2, -1, -2) if seed % 2 else (2, -3, -3), None)
        cfg.min_seed_length = min(k, 8)
        queries = []
        for _ in range(25):
            a, b = rng.randrange(0, 1400), rng.randrange(0, 1400)
            la, lb = rng.randrange(25, 60), rng.randrange(25, 60)
            mid = rand_seq(rng, rng.choice([0, 0, 1, 3, 8]))
            q = genome[a:a + la] + mid + genome[b:b + lb]
            if rng.random() < 0.5:
                q = mutate(rng, q)
            if rng.random() < 0.4:
                q = rc(q)
            queries.append(q)
        want = orc.AlignRun(g, cfg, queries).results()
        plain, got = product_chain(g, k, cfg, queries)
        for q in range(len(queries)):
            if got[q] is None:
                n_capacity += 1
                continue
            assert got[q] == want[q], (seed, q, queries[q], plain[q], got[q], want[q])
            n_chained += any(0 in a["nodes"] for a in got[q])
    assert n_chained > 20 and n_capacity <= 6, (n_chained, n_capacity)

"""Post-alignment chaining on the GPU path (config.post_chain_alignments = 1): the kernels keep every alignment of a query,
mgx_align_batch chains them on the host (metagraph_amd/csrc/chain_host.hpp).  The reference's 12 chain tests
(tests/graph/test_aligner_chain.cpp:36-269) and random stitched reads, complete alignment lists against the oracle."""
import random

import pytest

import orc
from metagraph_amd import aligner, capi
from test_oracle_chain import CASES, chain_config, chain_graph

pytestmark = pytest.mark.gpu


def gpu_align(g, k, cfg, queries):
    W, last, F, valid = g.export()
    A = aligner.Aligner(aligner.Graph(k, W, last, F, valid), cfg)
    got, status = A.align_batch(queries)
    assert A.stats()["extend_kernels"] & (capi.KERNEL_GRP8_ALT | capi.KERNEL_EXT64), A.stats()
    return got, status


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_reference_chain_tests_on_gpu(case):
    name, k, refs, query, scores, gaps, want, is_chain = case
    g = chain_graph(k, refs)
    cfg = chain_config(k, scores, gaps)
    (want_paths,) = orc.AlignRun(g, cfg, [query]).results()
    got, status = gpu_align(g, k, cfg, [query])
    assert status == [0]
    assert got[0] == want_paths
    assert len(got[0]) == 1 and got[0][0]["sequence"] == want and (0 in got[0][0]["nodes"]) == is_chain


def test_chaining_stitched_reads_on_gpu():
    from test_emu_vs_oracle import rand_seq, mutate, rc
    n_chained = n_capacity = 0
    for seed in range(4):
        rng = random.Random(900 + seed)
        k = rng.choice([12, 15, 21])
        genome = rand_seq(rng, 6000)
        g = orc.Graph.build(k, [genome], 0, False)
        cfg = chain_config(k, (2, -1, -2) if seed % 2 else (2, -3, -3), None)
        cfg.min_seed_length = k if seed % 2 else 10         # (many sub-k seeds: many alignments to keep)
        queries = []
        for _ in range(300):
            a, b = rng.randrange(0, 5800), rng.randrange(0, 5800)
            la, lb = rng.randrange(25, 90), rng.randrange(25, 90)
            q = genome[a:a + la] + rand_seq(rng, rng.choice([0, 0, 1, 3, 8])) + genome[b:b + lb]
            if rng.random() < 0.5:
                q = mutate(rng, q)
            if rng.random() < 0.4:
                q = rc(q)
            queries.append(q)
        want = orc.AlignRun(g, cfg, queries).results()
        got, status = gpu_align(g, k, cfg, queries)
        for q in range(len(queries)):
            if status[q]:
                assert status[q] == capi.MGX_ERR_CAPACITY and got[q] == []
                n_capacity += 1
                continue
            assert got[q] == want[q], (seed, q, queries[q], got[q], want[q])
            n_chained += any(0 in a["nodes"] for a in got[q])
    # (a query with more than 13 alignments above the cut-off has a capacity status: include/mgx.h, post_chain_alignments)
    assert n_chained > 100 and n_capacity <= 240, (n_chained, n_capacity)


def test_post_chain_with_an_annotation_is_refused():
    from labeled_worlds import labeled_world
    g, anno, reads = labeled_world(11, 15, n_strains=3, n_reads=4)
    W, last, F, valid = g.export()
    cfg = capi.config_cli(15)
    cfg.post_chain_alignments = 1
    with pytest.raises(Exception):
        aligner.Aligner(aligner.Graph(15, W, last, F, valid), cfg,
                        annotation=aligner.Annotation(g.n_edges, [anno.column_words(j) for j in range(anno.n_labels)]))

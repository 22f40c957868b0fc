"""Label-aware alignment (LabeledAligner / LabeledExtender / AnnotationBuffer) on the oracle, pinned by the
reference's own tests: tests/annotation/test_aligner_labeled.cpp — SimpleLinearGraph (:64-107), SimpleTangleGraph (:109-156),
SimpleTangleGraphSuffixSeed (:662-717) and, for a PRIMARY graph seen through CanonicalDBG, CanonicalTangleGraph (:719-778).
Expectations are transcribed by hand: per query, {label -> sequence of the alignment that must carry that label}.  The checks
are the reference's: as many alignments as expected labels; every alignment carries a label whose expected sequence is the
alignment's sequence; every reported label fully covers the alignment's sequence (get_alignment_labels :27-50)."""
import pytest

import orc
from metagraph_amd import capi

CASES = {
    "SimpleLinearGraph": dict(k=4, sequences=["GCAAT", "AATGCTT"], labels="AB", mode=0,
                              cfg=dict(max_seed_length=2 ** 64 - 1), matrix=("dna", 2, -1, -1),
                              expect={"GCAATGCTT": {"B": "AATGCTT", "A": "GCAAT"}}),
    "SimpleTangleGraph": dict(k=3, sequences=["TGCCT", "CGAATGCCT", "GGAATGCAT"], labels="ABC", mode=0,
                              cfg={}, matrix=("dna", 2, -1, -1),
                              expect={"CGAATGCAT": {"C": "GAATGCAT", "B": "CGAATGCCT", "A": "TGCCT"}}),
    "SimpleTangleGraphSuffixSeed": dict(k=4, sequences=["TGCCT", "TCGAATGCCT", "TGGAATGCAT"], labels="ABC", mode=0,
                                        cfg=dict(min_seed_length=2, left_end_bonus=5, right_end_bonus=5), matrix=("dna", 2, -1, -1),
                                        expect={"TGAAATGCAT": {"C": "TGGAATGCAT", "B": "TCGAATGCCT"}}),
    # :719-778: both halves of the test — a CANONICAL-mode DBGSuccinct (labels by the k-mer's representative: AnnotationBuffer's
    # spell_path + map_to_nodes branch, annotation_buffer.cpp:56-62,96-135) and a PRIMARY one seen through CanonicalDBG
    "CanonicalTangleGraph_canonical": dict(k=5, sequences=["GTCGAAA", "TTAGTCGAAA", "TCAGTCGATT"], labels="ABC", mode=1,
                                           cfg={}, matrix=("dna", 2, -1, -2),
                                           expect={"TTAGTTCAAA": {"B": "TTAGTCGAAA"}}),
    "CanonicalTangleGraph_primary": dict(k=5, sequences=["GTCGAAA", "TTAGTCGAAA", "TCAGTCGATT"], labels="ABC", mode=2,
                                         cfg={}, matrix=("dna", 2, -1, -2),
                                         expect={"TTAGTTCAAA": {"B": "TTAGTCGAAA"}}),
}


def build(case):
    if case["mode"] == 2:
        from test_oracle_primary_goldens import primary_contigs
        contigs = primary_contigs(case["sequences"], case["k"], "input")[0]
        g = orc.Graph.build(case["k"], contigs, 2, True)
        anno = orc.Annotation(g, len(case["labels"]))
        for j, seq in enumerate(case["sequences"]):
            anno.annotate(seq, j)
        return g, anno, orc.make_config(dict(case["cfg"]), case["matrix"])
    g = orc.Graph.build(case["k"], case["sequences"], case["mode"], True)        # build_graph_batch: dummy k-mers masked
    anno = orc.Annotation(g, len(case["labels"]))
    for j, seq in enumerate(case["sequences"]):
        anno.annotate(seq, j)
    cfg = orc.make_config(dict(case["cfg"]), case["matrix"])
    return g, anno, cfg


@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_labeled_kats(name):
    case = CASES[name]
    g, anno, cfg = build(case)
    import ctypes as C
    orc.L().orc_unfetched_label_lookups.restype = C.c_uint64
    before = orc.L().orc_unfetched_label_lookups()
    for query, want in case["expect"].items():
        run = orc.LabeledAlignRun(g, cfg, anno, [query])
        assert run.error == "", run.error
        # (the upstream cases stay inside what the reference defines: no label look-up of a node its buffer never fetched)
        assert orc.L().orc_unfetched_label_lookups() == before
        alns = run.results()[0]
        labs = run.labels()[0]
        assert len(alns) == len(want), (name, query, [(a["sequence"], a["cigar"], l) for a, l in zip(alns, labs)])
        for a, ls in zip(alns, labs):
            names = [case["labels"][l] for l in ls]
            assert names, a
            assert all(nm in want for nm in names), (names, want)
            assert any(want[nm] == a["sequence"] for nm in names), (a["sequence"], names, want)
            # every reported label covers every k-mer of the alignment's spelling (get_alignment_labels, check_full_coverage)
            n = g.n_edges
            if case["mode"] == 1:
                continue                                                          # (rows are those of the k-mers' representatives)
            rows = [(v - n if v > n else v) - 1 for v in a["nodes"] if v]         # (wrapper ids above n: the base node's row)
            for per_node in anno.get_rows(rows):
                assert set(ls) <= set(per_node)


def test_get_rows_is_the_column_major_bit_test():
    g, anno, _ = build(CASES["SimpleTangleGraph"])
    n = g.n_edges
    rows = list(range(n))
    got = anno.get_rows(rows)
    for j in range(3):
        words = anno.column_words(j)
        for r in rows:
            assert ((int(words[r >> 6]) >> (r & 63)) & 1) == (j in got[r])


# ---- self-consistency pins of the labelled restatement (the upstream label tests assert count / label / spelling only) ----
def _all_rows_label(g, n_labels=1):
    import ctypes as C
    anno = orc.Annotation(g, n_labels)
    orc.L().orc_annotation_set.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
    for r in range(g.n_edges):
        orc.L().orc_annotation_set(anno.h, r, 0)
    return anno


def test_one_label_on_every_node_is_the_unlabelled_aligner_on_the_kats():
    """With one label on every node nothing is ever filtered, flushed away or split in backtracking: LabeledAligner must give
    what DBGAligner gives with the seed lengths LabeledAligner's ctor clamps to (<= k) — scores, CIGARs, node ids — on every
    known-answer case with one alignment per query (with N > 1 the labelled backtracking stops once the seed's labels are
    accounted for, i.e. after one alignment per extension, by design: aligner_labeled.hpp:49-52)."""
    from test_oracle_kats import KATS
    n = 0
    for case in KATS["unit"]:
        if case["expect"].get("throws"):
            continue
        g = orc.Graph.build(case["k"], case["graph"], 0, case["mask_dummy"])
        cfg = orc.make_config(case["config"], case["matrix"])
        if cfg.num_alternative_paths != 1:
            continue
        k = case["k"]
        clamped = orc.make_config(case["config"], case["matrix"])
        clamped.min_seed_length = min(k, cfg.min_seed_length or k)
        clamped.max_seed_length = min(k, cfg.max_seed_length or k)
        u = orc.AlignRun(g, clamped, [case["query"]])
        l = orc.LabeledAlignRun(g, cfg, _all_rows_label(g), [case["query"]], validate=False)
        assert u.error == "" and l.error == "", (case["name"], u.error, l.error)
        assert l.results()[0] == u.results()[0], case["name"]
        assert all(ls == [0] for ls in l.labels()[0]), case["name"]
        n += 1
    assert n >= 45


def test_labels_as_connected_components_give_the_per_component_alignments():
    """Two unrelated genomes in one graph, a label each: labels never mix along a path, so an alignment that carries label c
    must be the alignment DBGAligner finds on component c's own graph (same score, CIGAR, spelling, strand; node ids differ
    between the graphs), and the component that aligns best is always reported."""
    import random
    from test_emu_vs_oracle import rand_seq, mutate, rc
    rng = random.Random(77)
    k = 15
    genomes = [rand_seq(rng, 1500), rand_seq(rng, 1500)]
    g = orc.Graph.build(k, genomes, 0, True)
    parts = [orc.Graph.build(k, [gn], 0, True) for gn in genomes]
    anno = orc.Annotation(g, 2)
    for j, gn in enumerate(genomes):
        anno.annotate(gn, j)
    cfg = capi.config_cli(k)
    cfg.min_seed_length = 11
    reads = []
    for i in range(60):
        src = genomes[i % 2]
        p = rng.randrange(0, len(src) - 100)
        r = mutate(rng, src[p:p + 100], rng.choice([0.0, 0.03]))
        reads.append(rc(r) if rng.random() < 0.5 else r)
    lab = orc.LabeledAlignRun(g, cfg, anno, reads)
    assert lab.error == ""
    per = [orc.AlignRun(pg, cfg, reads).results() for pg in parts]
    key = lambda a: (a["score"], a["cigar"], a["sequence"], a["orientation"], a["offset"])
    n_checked = 0
    for q, (alns, labs) in enumerate(zip(lab.results(), lab.labels())):
        best = max([per[c][q][0]["score"] for c in (0, 1) if per[c][q]], default=None)
        for a, ls in zip(alns, labs):
            assert len(ls) == 1, (q, ls)
            c = ls[0]
            assert per[c][q] and key(per[c][q][0]) == key(a), (q, c, a, per[c][q])
            n_checked += 1
        if best is not None:
            assert alns and alns[0]["score"] == best, (q, best, alns)
    assert n_checked >= 50

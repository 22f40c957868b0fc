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
    # :719-778, the PRIMARY half (a PRIMARY DBGSuccinct seen through CanonicalDBG; the CANONICAL-mode half needs the
    # AnnotationBuffer's spell_path + map_to_nodes branch, which is not restated)
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
    for query, want in case["expect"].items():
        run = orc.LabeledAlignRun(g, cfg, anno, [query])
        assert run.error == "", run.error
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

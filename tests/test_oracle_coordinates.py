"""Seed chaining + label coordinates (the remainder) on the oracle: LabeledAligner with an annotation that carries
k-mer coordinates switches chain_alignments on and the global x-drop off (aligner_labeled.cpp:457-462), so these cases run
call_seed_chains_both_strands / chain_seeds (aligner_chainer.cpp:64-542), extend_chain / align_connect (dbg_aligner.cpp:155-250,
388-529), the coordinate-consistent LabeledExtender (aligner_labeled.cpp:245-300,361-448) and Alignment::splice* together.
Pinned by the reference's own tests, transcribed as data in tests/golden/aligner_coordinate_kats.json
(tests/annotation/test_aligner_labeled.cpp:158-258,319-471,604-660); the checks are the reference's."""
import json
import os

import pytest

import orc

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "aligner_coordinate_kats.json")))
MATRIX = ("dna", 2, -1, -1)


def build(k, sequences, label_of, coord_starts=None):
    g = orc.Graph.build(k, sequences, 0, True)                   # build_graph_batch<DBGSuccinct>: dummy k-mers masked
    anno = orc.Annotation(g, max(label_of) + 1)
    for i, seq in enumerate(sequences):
        anno.annotate_coords(seq, label_of[i], coord_starts[i] if coord_starts else 0)
    return g, anno


@pytest.mark.parametrize("case", KATS["tangle"], ids=lambda c: c["name"])
def test_tangle_graphs_with_coordinates(case):
    """LabeledAlignerTest.SimpleTangleGraphCoords / ...Middle / ...Cycle: one alignment per expected label; every alignment carries
    a label whose expected spelling is the alignment's, with that label's first coordinate as expected."""
    g, anno = build(case["k"], case["sequences"], list(range(len(case["labels"]))))
    cfg = orc.make_config({}, MATRIX)
    run = orc.LabeledAlignRun(g, cfg, anno, [case["query"]])
    assert run.error == "", run.error
    alns, labs, coords = run.results()[0], run.labels()[0], run.coordinates()[0]
    want = case["expect"]
    got = [(a["sequence"], a["cigar"], [case["labels"][l] for l in ls], cs) for a, ls, cs in zip(alns, labs, coords)]
    assert len(alns) == len(want), got
    for a, ls, cs in zip(alns, labs, coords):
        assert len(ls) == len(cs) and all(len(c) > 0 for c in cs), got
        found = False
        for l, c in zip(ls, cs):
            name = case["labels"][l]
            assert name in want, got
            if a["sequence"] == want[name][0]:
                found = True
                assert c[0] == want[name][1], got
                if name in case.get("cigar_comments", {}):           # (the CIGAR the reference test notes in a comment)
                    assert a["cigar"] == case["cigar_comments"][name], got
                break
        assert found, got


@pytest.mark.parametrize("case", KATS["coord"], ids=lambda c: c["name"])
def test_coordinates_across_sequence_boundaries(case):
    """LabeledAlignerCoordTest (k = 5, one label for all sequences, coordinate offsets per sequence, max_seed_length unbounded):
    exactly one alignment, with its CIGAR, its coordinate list and Alignment::format_coords(CoordToHeader(...), k)."""
    k = 5
    g, anno = build(k, case["sequences"], [0] * len(case["sequences"]), case["coord_starts"])
    over = {"max_seed_length": 2 ** 64 - 1}
    if "min_exact_match" in case:
        over["min_exact_match"] = case["min_exact_match"]
    cfg = orc.make_config(over, MATRIX)
    run = orc.LabeledAlignRun(g, cfg, anno, [case["query"]])
    assert run.error == "", run.error
    alns, labs, coords = run.results()[0], run.labels()[0], run.coordinates()[0]
    got = [(a["sequence"], a["cigar"], ls, cs) for a, ls, cs in zip(alns, labs, coords)]
    assert len(alns) == 1, got
    a = alns[0]
    assert a["cigar"] == case["cigar"], got
    if "orientation" in case:
        assert int(a["orientation"]) == case["orientation"], got
    if "sequence" in case:
        assert a["sequence"] == case["sequence"], got
    if "sequence_length" in case:
        assert len(a["sequence"]) == case["sequence_length"], got
    assert len(coords[0]) == 1 and coords[0][0] == case["coordinates"], got
    assert run.format_coords(0, 0, [case["headers"]], [case["kmer_counts"]], k) == case["format_coords"], got

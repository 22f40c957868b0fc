"""Label-aware alignment in the kernels' wave program (label_sets.hpp / label_driver.hpp: LabeledExtender's flush /
call_outgoing / label-bounded backtracking, the per-label aggregator, filter_seeds) under the host model, against the
oracle's LabeledAligner — full alignment lists AND their label sets: the reference's label tests on BASIC graphs
(tests/annotation/test_aligner_labeled.cpp, see test_oracle_labeled.py) and random worlds of diverged strains.  CPU only; the
GPU half is tests/test_gpu_labels.py."""
import pytest

import emu_drv
import orc
from metagraph_amd import capi
from labeled_worlds import labeled_world, with_labels
from test_oracle_labeled import CASES, build


SEEN = {"multi_aln": 0, "multi_label": 0, "worlds": 0}


def compare_emu_labeled(g, anno, cfg, reads, validate=True, mode=0):
    o = orc.LabeledAlignRun(g, cfg, anno, reads, validate=validate)
    assert o.error == "", o.error
    e = emu_drv.EmuRun(emu_drv.EmuGraph(g, mode=mode), cfg, reads, annotation=emu_drv.EmuAnnotation(anno))
    assert e.error == "", e.error
    got, status = e.results()
    want = with_labels(o)
    assert all(s == 0 for s in status), status
    for q in range(len(reads)):
        assert got[q] == want[q], (q, reads[q], got[q], want[q])
    info = e.seed_info()                                   # the label filter's products: seed lists and num_matching per strand
    for strand in (0, 1):
        for q, (ss, nm) in enumerate(o.seeds(strand)):
            assert info[q]["num_matches"][strand] == nm, (q, strand)
            assert info[q]["seeds"][strand] == emu_drv.oracle_seeds_as_tuples(ss), (q, strand, reads[q])
    return want


@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_label_kats_in_the_wave_program(name):
    case = CASES[name]
    g, anno, cfg = build(case)
    for query, expect in case["expect"].items():
        want = compare_emu_labeled(g, anno, cfg, [query], validate=not (cfg.left_end_bonus or cfg.right_end_bonus), mode=case["mode"])
        assert len(want[0]) == len(expect)


@pytest.mark.parametrize("seed,k,n_strains,divergence", [(1, 11, 3, 0.02), (2, 19, 6, 0.05), (3, 7, 2, 0.02), (4, 31, 4, 0.01),
                                                        (5, 12, 1, 0.0), (6, 15, 6, 0.02)])
def test_random_labeled_worlds(seed, k, n_strains, divergence, monkeypatch):
    g, anno, reads = labeled_world(seed, k, n_strains=n_strains, divergence=divergence)
    cfg = capi.config_cli(k)
    if seed % 2:
        cfg.min_seed_length = max(5, k - 4)                # sub-k seeds
    if seed == 6:
        cfg.num_alternative_paths = 2
    if seed % 3 == 0:
        monkeypatch.setenv("MGX_EMU_SPLIT", "1")           # seeding phase, sort, extension phase (the device's pipeline)
    want = compare_emu_labeled(g, anno, cfg, reads)
    SEEN["multi_aln"] += sum(1 for a in want if len(a) > 1)
    SEEN["multi_label"] += sum(1 for a in want for x in a if len(x["labels"]) > 1)
    SEEN["worlds"] += 1
    if SEEN["worlds"] == 6:          # (over the six worlds: reads with several alignments and alignments with several labels occur)
        assert SEEN["multi_aln"] >= 5 and SEEN["multi_label"] >= 5, SEEN


def paralog_world(seed, k, copies=4):
    """one labelled genome holding several diverged copies of a segment: a read from one copy aligns to all of them, under the
    same label — what num_alternative_paths > 1 is for"""
    import random
    from test_emu_vs_oracle import rand_seq, mutate, rc
    rng = random.Random(seed)
    seg = rand_seq(rng, 160)
    parts = []
    for c in range(copies):
        parts.append(rand_seq(rng, 60))
        parts.append(seg if c == 0 else mutate(rng, seg, 0.04))
    genome = "".join(parts) + rand_seq(rng, 60)
    g = orc.Graph.build(k, [genome], 0, False)
    anno = orc.Annotation(g, 2)
    anno.annotate(genome, 0)
    anno.annotate(genome[:len(genome) // 2], 1)
    reads = []
    for i in range(12):
        a = rng.randrange(0, 160 - 90)
        r = mutate(rng, seg[a:a + 90], rng.choice([0.0, 0.02]))
        reads.append(rc(r) if i % 3 == 0 else r)
    return g, anno, reads


@pytest.mark.parametrize("seed,k,n_alt", [(7, 15, 4), (8, 11, 3), (9, 12, 4)])
def test_labeled_worlds_with_up_to_four_alignments_per_label(seed, k, n_alt):
    g, anno, reads = paralog_world(seed, k)
    cfg = capi.config_cli(k)
    cfg.num_alternative_paths = n_alt
    cfg.rel_score_cutoff = 0.0                         # (keep the weaker copies' alignments)
    want = compare_emu_labeled(g, anno, cfg, reads)
    assert any(len(a) > 2 for a in want), [len(a) for a in want]


def primary_labeled_world(seed, k):
    """a PRIMARY graph (seen through the CanonicalDBG wrapper) over two strains, labels per strain and per segment"""
    import random
    from test_emu_vs_oracle import rand_seq, mutate, rc
    from test_oracle_primary_goldens import primary_contigs
    rng = random.Random(seed)
    genome = rand_seq(rng, 900)
    strains = [genome, mutate(rng, genome, 0.03)]
    contigs = primary_contigs(strains, k, "input")[0]
    g = orc.Graph.build(k, contigs, 2, True)
    anno = orc.Annotation(g, 4)
    for j, st in enumerate(strains):
        anno.annotate(st, j)
    anno.annotate(genome[100:400], 2)
    anno.annotate(rc(genome[500:800]), 3)                 # (annotated from the other strand: the same base nodes)
    reads = []
    for i in range(16):
        st = rng.choice(strains)
        p = rng.randrange(0, len(st) - 80)
        r = mutate(rng, st[p:p + 80], rng.choice([0.0, 0.03]))
        reads.append(rc(r) if i % 2 else r)
    return g, anno, reads


@pytest.mark.parametrize("seed,k", [(1, 11), (2, 15), (3, 7)])
def test_labeled_alignment_on_primary_graphs(seed, k):
    g, anno, reads = primary_labeled_world(seed, k)
    cfg = capi.config_cli(k)
    if seed == 2:
        cfg.min_seed_length = 11
    want = compare_emu_labeled(g, anno, cfg, reads, mode=2)
    assert sum(1 for a in want if a) >= 8


def canonical_labeled_world(seed, k):
    """a CANONICAL-mode graph (both strands stored) over two strains; labels live on the k-mers' representatives"""
    import random
    from test_emu_vs_oracle import rand_seq, mutate, rc
    rng = random.Random(seed)
    genome = rand_seq(rng, 700)
    strains = [genome, mutate(rng, genome, 0.03)]
    g = orc.Graph.build(k, strains, 1, True)
    anno = orc.Annotation(g, 4)
    for j, st in enumerate(strains):
        anno.annotate(st, j)
    anno.annotate(genome[100:350], 2)
    anno.annotate(rc(genome[400:650]), 3)                 # (annotated from the other strand: the same representatives)
    reads = []
    for i in range(16):
        st = rng.choice(strains)
        p = rng.randrange(0, len(st) - 80)
        r = mutate(rng, st[p:p + 80], rng.choice([0.0, 0.03]))
        reads.append(rc(r) if i % 2 else r)
    return g, anno, reads


@pytest.mark.parametrize("seed,k", [(1, 11), (2, 15), (3, 7)])
def test_labeled_alignment_on_canonical_mode_graphs(seed, k):
    g, anno, reads = canonical_labeled_world(seed, k)
    cfg = capi.config_cli(k)
    if seed == 2:
        cfg.min_seed_length = 11
    want = compare_emu_labeled(g, anno, cfg, reads, mode=1)
    assert sum(1 for a in want if a) >= 8


def many_labels_world(n_labels=100, k=11, seed=77):
    """one genome carrying n_labels labels — more than the label arenas of a first run hold (64 queues / labels per read)"""
    import random
    from test_emu_vs_oracle import rand_seq, mutate
    rng = random.Random(seed)
    genome = rand_seq(rng, 500)
    g = orc.Graph.build(k, [genome], 0, False)
    anno = orc.Annotation(g, n_labels)
    for j in range(n_labels):
        anno.annotate(genome if j % 3 else genome[50:450], j)        # (two kinds of rows)
    reads = [mutate(rng, genome[a:a + 80], 0.02) for a in (10, 120, 300, 400)]
    return g, anno, reads


def test_label_arenas_grow_with_the_capacity_retry(monkeypatch):
    """A read with more labels than the first run's arenas hold reports a capacity status there — never a wrong result — and is
    exact with the arenas the retry of mgx_align_batch derives (derive_limits' label_scale)."""
    g, anno, reads = many_labels_world()
    cfg = capi.config_cli(11)
    e = emu_drv.EmuRun(emu_drv.EmuGraph(g), cfg, reads, annotation=emu_drv.EmuAnnotation(anno))
    assert e.error == "", e.error
    _, status = e.results()
    assert any(s == capi.MGX_ERR_CAPACITY for s in status) and all(s in (0, capi.MGX_ERR_CAPACITY) for s in status), status
    monkeypatch.setenv("MGX_EMU_LABEL_SCALE", "2")
    want = compare_emu_labeled(g, anno, cfg, reads)
    assert max(len(x["labels"]) for a in want for x in a) > 64

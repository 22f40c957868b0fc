"""The device pieces of seed chaining / label coordinates against the oracle, whose whole chaining path is
pinned by the reference's nine coordinate KATs (tests/test_oracle_coordinates.py):
  * mgx_chain_seeds = chain_seeds (aligner_chainer.cpp:341-542): the sort of the anchors and the banded DP incl. its float gap cost;
  * mgx_annotation_set_coordinates / mgx_annotation_get_row_tuples = MultiIntMatrix::get_row_tuples over a ColumnCoordAnnotator.
LabeledAligner's coordinate mode as a whole is NOT on the device: an annotation with coordinates is refused at aligner creation."""
import ctypes as C
import random

import numpy as np
import pytest

import orc
from metagraph_amd import aligner, capi

pytestmark = pytest.mark.gpu


class Anchor(C.Structure):
    _fields_ = [("label", C.c_uint64), ("coordinate", C.c_int64), ("seed_clipping", C.c_int32), ("seed_end", C.c_int32),
                ("chain_score", C.c_int32), ("seed_index", C.c_uint32)]


def random_lists(rng, n_lists, max_anchors, query_size, n_labels, coord_span):
    """anchor lists like a read's: seeds along the query on a few loci per label (+ noise), unique sort keys"""
    lists = []
    for _ in range(n_lists):
        n = rng.randrange(0, max_anchors + 1)
        seen, out = set(), []
        loci = [(rng.randrange(n_labels), rng.randrange(coord_span)) for _ in range(rng.randrange(1, 5))]
        while len(out) < n:
            lab, base = rng.choice(loci)
            clip = rng.randrange(0, query_size - 4)
            length = rng.randrange(3, min(32, query_size - clip) + 1)
            coord = base + clip + rng.choice([0, 0, 0, 0, 1, -1, 2, -3, rng.randrange(-40, 40)])
            if rng.random() < 0.1:
                coord = rng.randrange(coord_span)
            key = (lab, coord, clip, clip + length)
            if key in seen:
                continue
            seen.add(key)
            out.append((lab, coord, clip, clip + length, length, len(out)))
        lists.append(out)
    return lists


def run_both(cfg, lists, query_sizes):
    n = sum(len(l) for l in lists)
    begin = np.zeros(len(lists) + 1, dtype=np.uint64)
    begin[1:] = np.cumsum([len(l) for l in lists])
    flat = (Anchor * max(1, n))()
    x = 0
    for l in lists:
        for a in l:
            flat[x] = Anchor(*a)
            x += 1
    qs = np.array(query_sizes, dtype=np.uint32)
    lib = capi.lib()
    lib.mgx_chain_seeds.argtypes = [C.POINTER(capi.Config), C.c_int, C.POINTER(Anchor), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                    C.c_uint64, C.POINTER(Anchor), C.POINTER(C.c_uint32)]
    g_sorted = (Anchor * max(1, n))()
    g_back = (C.c_uint32 * max(1, n))()
    rc = lib.mgx_chain_seeds(C.byref(cfg), 0, flat, begin.ctypes.data_as(C.POINTER(C.c_uint64)), qs.ctypes.data_as(C.POINTER(C.c_uint32)),
                             len(lists), g_sorted, g_back)
    assert rc == 0, lib.mgx_last_error()
    o = (Anchor * max(1, n))(*flat)
    o_back = (C.c_uint32 * max(1, n))()
    ol = orc.L()
    ol.orc_chain_seeds.argtypes = [C.POINTER(capi.Config), C.POINTER(Anchor), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_uint64,
                                   C.POINTER(C.c_uint32)]
    ol.orc_chain_seeds(C.byref(cfg), o, begin.ctypes.data_as(C.POINTER(C.c_uint64)), qs.ctypes.data_as(C.POINTER(C.c_uint32)), len(lists), o_back)
    tup = lambda a: (a.label, a.coordinate, a.seed_clipping, a.seed_end, a.chain_score, a.seed_index)
    return [tup(g_sorted[i]) for i in range(n)], [g_back[i] for i in range(n)], [tup(o[i]) for i in range(n)], [o_back[i] for i in range(n)]


@pytest.mark.parametrize("seed,query_size,min_seed_length", [(1, 150, 19), (2, 150, 31), (3, 60, 5), (4, 1000, 15), (5, 250, 100)])
def test_chain_seeds_matches_the_oracle(seed, query_size, min_seed_length):
    rng = random.Random(seed)
    cfg = capi.config_cli(31)
    cfg.min_seed_length = min_seed_length
    lists = random_lists(rng, 300, 400, query_size, 4, 5000)
    lists += [[], [(0, 10, 5, 20, 15, 0)]]                       # empty list, a single anchor
    gs, gb, os_, ob = run_both(cfg, lists, [query_size] * len(lists))
    assert gs == os_
    assert gb == ob
    assert any(b != 0xFFFFFFFF for b in ob)                      # (chains were found)


def test_chain_seeds_gap_costs_cover_every_coordinate_difference():
    """two anchors per list, one list per coordinate difference: the float gap cost of every difference a 2000-bp query allows"""
    cfg = capi.config_cli(31)
    lists, qs = [], []
    for d in range(0, 1990):
        # anchor i (larger coordinate, larger clipping) precedes anchor j: dist = 5, coord_dist = 5 + d
        lists.append([(0, 100000 + 5 + d, 1005, 1036, 31, 0), (0, 100000, 1000, 1031, 31, 1)])
        qs.append(2000)
    gs, gb, os_, ob = run_both(cfg, lists, qs)
    assert gs == os_ and gb == ob


def test_long_lists_and_many_labels():
    rng = random.Random(99)
    cfg = capi.config_cli(31)
    lists = random_lists(rng, 6, 6000, 5000, 40, 200000)
    gs, gb, os_, ob = run_both(cfg, lists, [5000] * len(lists))
    assert gs == os_ and gb == ob


def test_row_tuples_on_the_device_and_the_refusal_of_coordinate_mode():
    """MultiIntMatrix::get_row_tuples: labels and coordinates of a batch of rows from the device's row-major copy equal the
    oracle's Annotation (annotate_kmer_coords over three overlapping sequences under two labels)."""
    rng = random.Random(5)
    k = 7
    genome = "".join(rng.choice("ACGT") for _ in range(600))
    seqs = [genome[:400], genome[200:], genome[100:300] + genome[100:300]]        # overlaps and a repeat: several coordinates per k-mer
    labs = [0, 1, 0]
    g = orc.Graph.build(k, seqs, 0, True)
    anno = orc.Annotation(g, 2)
    for s, l, st in zip(seqs, labs, [0, 1000, 5000]):
        anno.annotate_coords(s, l, st)
    n_rows = g.n_edges
    # the sparse form from the oracle's own matrix + coordinates (through its get_row_tuples-shaped accessors)
    ol = orc.L()
    ol.orc_annotation_row_tuples.restype = C.c_uint64
    ol.orc_annotation_row_tuples.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.c_uint64]
    per_row = []
    for r in range(n_rows):
        labels = (C.c_uint64 * 8)()
        cb = (C.c_uint64 * 9)()
        co = (C.c_int64 * 64)()
        nl = ol.orc_annotation_row_tuples(anno.h, r, labels, cb, co, 64)
        per_row.append([(int(labels[t]), [int(co[x]) for x in range(cb[t], cb[t + 1])]) for t in range(nl)])
    cols = {0: [], 1: []}
    for r, tuples in enumerate(per_row):
        for lab, coords in tuples:
            cols[lab].append((r, coords))
    col_begin, rows, coord_begin, coords = [0], [], [0], []
    for lab in (0, 1):
        rng.shuffle(cols[lab])                                    # (any order within a column)
        for r, cs in cols[lab]:
            rows.append(r)
            coords += cs
            coord_begin.append(len(coords))
        col_begin.append(len(rows))
    lib = capi.lib()
    A = C.c_void_p()
    cbv = (C.c_uint64 * len(col_begin))(*col_begin)
    rwv = (C.c_uint64 * max(1, len(rows)))(*rows)
    # (no argtypes on entry points other tests call with raw addresses: the library object is shared by the session)
    assert lib.mgx_annotation_create_sparse(C.c_uint64(n_rows), C.c_uint32(2), C.cast(cbv, C.c_void_p), C.cast(rwv, C.c_void_p), 0, 0, C.byref(A)) == 0, lib.mgx_last_error()
    lib.mgx_annotation_set_coordinates.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int64)]
    cob = (C.c_uint64 * len(coord_begin))(*coord_begin)
    cov = (C.c_int64 * max(1, len(coords)))(*coords)
    lib.mgx_annotation_has_coordinates.argtypes = [C.c_void_p]
    assert lib.mgx_annotation_has_coordinates(A) == 0
    assert lib.mgx_annotation_set_coordinates(A, 2, cbv, rwv, cob, cov) == 0, lib.mgx_last_error()
    assert lib.mgx_annotation_has_coordinates(A) == 1
    want_rows = [rng.randrange(n_rows) for _ in range(500)] + list(range(min(50, n_rows)))
    n = len(want_rows)
    rv = (C.c_uint64 * n)(*want_rows)
    ob = (C.c_uint64 * (n + 1))()
    olab = (C.c_uint32 * (4 * n))()
    ocb = (C.c_uint64 * (4 * n + 1))()
    oco = (C.c_int64 * (64 * n))()
    nl, nc = C.c_uint64(), C.c_uint64()
    lib.mgx_annotation_get_row_tuples.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                                  C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.c_uint64, C.POINTER(C.c_uint64),
                                                  C.POINTER(C.c_uint64)]
    assert lib.mgx_annotation_get_row_tuples(A, rv, n, ob, olab, 4 * n, ocb, oco, 64 * n, C.byref(nl), C.byref(nc)) == 0, lib.mgx_last_error()
    for i, r in enumerate(want_rows):
        got = [(int(olab[e]), [int(oco[x]) for x in range(ocb[e], ocb[e + 1])]) for e in range(ob[i], ob[i + 1])]
        assert got == per_row[r], (r, got, per_row[r])
    assert any(len(cs) > 1 for t in per_row for _, cs in t)      # (the repeat gave k-mers several coordinates)
    # LabeledAligner with such an annotation chains seeds (aligner_labeled.cpp:457-462): not on the device -> refused, not
    # answered in the plain label-aware mode
    W, last, F, valid = g.export()
    G = aligner.Graph(k, W, last, F, valid)
    out = C.c_void_p()
    cfg = capi.config_cli(k)
    assert lib.mgx_labeled_aligner_create(G.h, C.byref(cfg), None, A, C.byref(out)) == capi.MGX_ERR_UNSUPPORTED
    lib.mgx_annotation_destroy(A)

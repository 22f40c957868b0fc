"""The device label matrix (mgx_annotation_create / mgx_annotation_get_rows, metagraph_amd/csrc/mgx_annot.hip) against the
oracle's column-major bit test (oracle/orc_align.hpp struct Annotation = ColumnMajor::get_rows, column_major.cpp:27-44):
the reference's labeled test graphs, random sparse and dense matrices (rows without labels, with one — answered from the
head word alone — and with many), rows past the matrix, the capacity answer, device-resident input / output."""
import ctypes as C
import random

import numpy as np
import pytest

import orc
from metagraph_amd import capi

pytestmark = pytest.mark.gpu


def device_rows(h, rows, cap=None):
    L = capi.lib()
    n = len(rows)
    r = np.asarray(rows, dtype=np.uint64)
    begin = np.zeros(n + 1, dtype=np.uint64)
    need = C.c_uint64()
    if cap is None:
        rc = L.mgx_annotation_get_rows(h, r.ctypes.data, n, 0, begin.ctypes.data, None, 0, 0, C.byref(need))
        assert rc in (capi.MGX_OK, capi.MGX_ERR_CAPACITY), L.mgx_last_error()
        cap = int(need.value)
    labels = np.zeros(max(1, cap), dtype=np.uint32)
    rc = L.mgx_annotation_get_rows(h, r.ctypes.data, n, 0, begin.ctypes.data, labels.ctypes.data, cap, 0, C.byref(need))
    assert rc == capi.MGX_OK, L.mgx_last_error()
    return [[int(x) for x in labels[int(begin[i]):int(begin[i + 1])]] for i in range(n)]


def make_device(columns, n_rows):
    L = capi.lib()
    arrs = [np.ascontiguousarray(c, dtype=np.uint64) for c in columns]
    ptrs = (C.c_void_p * max(1, len(arrs)))(*[a.ctypes.data for a in arrs])
    h = C.c_void_p()
    rc = L.mgx_annotation_create(n_rows, len(arrs), ptrs, 0, C.byref(h))
    assert rc == capi.MGX_OK, L.mgx_last_error()
    return h, arrs


def test_get_rows_on_the_reference_label_graphs():
    from test_oracle_labeled import CASES, build
    L = capi.lib()
    for name, case in CASES.items():
        g, anno, _ = build(case)
        n = g.n_edges
        h, keep = make_device([anno.column_words(j) for j in range(len(case["labels"]))], n)
        rows = list(range(n)) + [0, n - 1, n + 5]          # every row, repeats, one past the matrix (no labels)
        want = anno.get_rows(list(range(n)) + [0, n - 1]) + [[]]
        assert device_rows(h, rows) == want, name
        L.mgx_annotation_destroy(h)


@pytest.mark.parametrize("n_rows,n_labels,density", [(1000, 3, 0.5), (5000, 40, 0.02), (70000, 1000, 0.001), (300, 70, 0.9)])
def test_get_rows_on_random_matrices(n_rows, n_labels, density):
    rng = np.random.default_rng(n_rows + n_labels)
    n_words = (n_rows + 63) // 64
    cols = []
    for j in range(n_labels):
        bits = rng.random(n_words * 64) < density
        bits[n_rows:] = False
        cols.append(np.packbits(bits.reshape(n_words, 64)[:, ::-1], axis=1).view(">u8").astype(np.uint64).reshape(n_words))
    h, keep = make_device(cols, n_rows)
    rows = [int(x) for x in rng.integers(0, n_rows, size=4000)]
    got = device_rows(h, rows)
    for r, labels in zip(rows, got):
        want = [j for j in range(n_labels) if (int(cols[j][r >> 6]) >> (r & 63)) & 1]
        assert labels == want
    # capacity: a buffer one entry short is refused, the needed size reported
    L = capi.lib()
    total = sum(len(x) for x in got)
    if total > 1:
        r = np.asarray(rows, dtype=np.uint64)
        begin = np.zeros(len(rows) + 1, dtype=np.uint64)
        labels = np.zeros(total, dtype=np.uint32)
        need = C.c_uint64()
        rc = L.mgx_annotation_get_rows(h, r.ctypes.data, len(rows), 0, begin.ctypes.data, labels.ctypes.data, total - 1, 0, C.byref(need))
        assert rc == capi.MGX_ERR_CAPACITY and need.value == total and int(begin[-1]) == total
    L.mgx_annotation_destroy(h)


@pytest.mark.parametrize("n_rows,n_labels,density", [(1000, 3, 0.5), (70000, 1000, 0.001), (300, 70, 0.9), (500, 5, 0.0)])
def test_sparse_construction_gives_the_same_matrix(n_rows, n_labels, density):
    """mgx_annotation_create_sparse (the columns' set rows: a ColumnCompressed annotation's content) == mgx_annotation_create
    (column bit vectors) on every row"""
    from metagraph_amd import aligner
    rng = np.random.default_rng(7 * n_rows + n_labels)
    cols, col_begin, rows_all = [], [0], []
    for j in range(n_labels):
        bits = rng.random(n_rows) < density
        w = np.zeros((n_rows + 63) // 64, dtype=np.uint64)
        r = np.nonzero(bits)[0].astype(np.int64)
        np.bitwise_or.at(w, r >> 6, np.uint64(1) << (r & 63).astype(np.uint64))
        cols.append(w)
        rr = r.copy()
        rng.shuffle(rr)                                   # (any order within a column)
        rows_all.append(rr.astype(np.uint64))
        col_begin.append(col_begin[-1] + len(rr))
    h, keep = make_device(cols, n_rows)
    S = aligner.Annotation.from_sparse(n_rows, np.asarray(col_begin, dtype=np.uint64),
                                       np.concatenate(rows_all) if rows_all else np.zeros(0, dtype=np.uint64))
    q = list(range(n_rows)) + [n_rows + 3]
    assert device_rows(S.h, q) == device_rows(h, q)
    capi.lib().mgx_annotation_destroy(h)

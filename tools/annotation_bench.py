"""BASELINE config 3's hot primitive on the device: get_rows over a 1000-label annotation of a 100 M-row matrix (labels =
1000 contiguous genome segments: every row has one label, rows next to a boundary two), for the seed nodes of a read batch
(10 M reads x ~4 seed nodes = 40 M random rows).  Prints rows/s and the achieved fraction of the random-line ceiling.
    python tools/annotation_bench.py [--rows 100000000 --labels 1000 --queries 40000000]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metagraph_amd import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=100_000_000)
ap.add_argument("--labels", type=int, default=1000)
ap.add_argument("--queries", type=int, default=40_000_000)
args = ap.parse_args()
L = capi.lib()
n_rows, n_labels = args.rows, args.labels
n_words = (n_rows + 63) // 64
seg = (n_rows + n_labels - 1) // n_labels
t0 = time.time()
cols = []
for j in range(n_labels):
    lo, hi = max(0, j * seg - 15), min(n_rows, (j + 1) * seg + 15)      # segments overlap by 30 rows: two labels at the boundaries
    w = np.zeros(n_words, dtype=np.uint64)
    if hi > lo:
        fw, lw = lo >> 6, (hi - 1) >> 6
        w[fw:lw + 1] = np.uint64(0xFFFFFFFFFFFFFFFF)
        w[fw] &= np.uint64((0xFFFFFFFFFFFFFFFF << (lo & 63)) & 0xFFFFFFFFFFFFFFFF)
        if hi & 63:
            w[lw] &= np.uint64((1 << (hi & 63)) - 1)
    cols.append(w)
t_host = time.time() - t0
ptrs = (C.c_void_p * n_labels)(*[c.ctypes.data for c in cols])
h = C.c_void_p()
t0 = time.time()
rc = L.mgx_annotation_create(n_rows, n_labels, ptrs, 0, C.byref(h))
assert rc == 0, L.mgx_last_error()
t_create = time.time() - t0
del cols
rng = np.random.default_rng(7)
rows = rng.integers(0, n_rows, size=args.queries, dtype=np.uint64)
begin = np.zeros(args.queries + 1, dtype=np.uint64)
labels = np.zeros(args.queries * 2 + 16, dtype=np.uint32)
need = C.c_uint64()
best = None
for rep in range(3):
    t0 = time.time()
    rc = L.mgx_annotation_get_rows(h, rows.ctypes.data, args.queries, 0, begin.ctypes.data, labels.ctypes.data, len(labels), 0, C.byref(need))
    dt = time.time() - t0
    assert rc == 0, L.mgx_last_error()
    best = dt if best is None else min(best, dt)
# spot check against the construction
for i in rng.integers(0, args.queries, size=2000):
    r = int(rows[i]); got = [int(x) for x in labels[int(begin[i]):int(begin[i + 1])]]
    want = [j for j in (r // seg - 1, r // seg, r // seg + 1) if 0 <= j < n_labels and max(0, j * seg - 15) <= r < min(n_rows, (j + 1) * seg + 15)]
    assert got == want, (r, got, want)
print(json.dumps({"rows": n_rows, "labels": n_labels, "device_bytes": int(L.mgx_annotation_device_bytes(h)),
                  "create_s_incl_h2d_of_columns": round(t_create, 2), "queries": args.queries,
                  "get_rows_s_host_in_host_out": round(best, 4), "rows_per_s": round(args.queries / best),
                  "labels_returned": int(need.value), "spot_check": "2000 rows ok"}))

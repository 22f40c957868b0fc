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
# device-resident I/O, timed with HIP events on the stream the kernels run on (the null stream): what the call costs when its
# caller keeps rows and results in HBM — no PCIe in the figure
hip = C.CDLL("libamdhip64.so")
def hip_ok(rc):
    assert rc == 0, "hip error %d" % rc
d_rows, d_begin, d_labels = C.c_void_p(), C.c_void_p(), C.c_void_p()
hip_ok(hip.hipMalloc(C.byref(d_rows), C.c_size_t(args.queries * 8)))
hip_ok(hip.hipMalloc(C.byref(d_begin), C.c_size_t((args.queries + 1) * 8)))
hip_ok(hip.hipMalloc(C.byref(d_labels), C.c_size_t(len(labels) * 4)))
hip_ok(hip.hipMemcpy(d_rows, C.c_void_p(rows.ctypes.data), C.c_size_t(args.queries * 8), 1))
ev0, ev1 = C.c_void_p(), C.c_void_p()
hip_ok(hip.hipEventCreate(C.byref(ev0))); hip_ok(hip.hipEventCreate(C.byref(ev1)))
dev_ms = []
for rep in range(5):
    hip_ok(hip.hipEventRecord(ev0, None))
    rc = L.mgx_annotation_get_rows(h, d_rows, args.queries, 1, d_begin, d_labels, len(labels), 1, C.byref(need))
    hip_ok(hip.hipEventRecord(ev1, None))
    hip_ok(hip.hipEventSynchronize(ev1))
    assert rc == 0, L.mgx_last_error()
    ms = C.c_float()
    hip_ok(hip.hipEventElapsedTime(C.byref(ms), ev0, ev1))
    dev_ms.append(ms.value)
begin_d = np.zeros(args.queries + 1, dtype=np.uint64)
labels_d = np.zeros(len(labels), dtype=np.uint32)
hip_ok(hip.hipMemcpy(C.c_void_p(begin_d.ctypes.data), d_begin, C.c_size_t((args.queries + 1) * 8), 2))
hip_ok(hip.hipMemcpy(C.c_void_p(labels_d.ctypes.data), d_labels, C.c_size_t(len(labels) * 4), 2))
assert np.array_equal(begin_d, begin) and np.array_equal(labels_d[:int(need.value)], labels[:int(need.value)]), "device-resident results differ"
dev_best = min(dev_ms[1:]) * 1e-3
# every row costs one dependent 8-byte access to head[] (a 64-byte line), rows with two labels one more to more[]
lines = args.queries + int((np.diff(begin) >= 2).sum())
ceiling = None
try:
    ceil = json.load(open(os.path.join(ROOT, "profiles", "r03_gather_ceiling.json")))
    ceiling = max(r["chains1"] for r in ceil["sets"][1]["rows"]) * 1e9       # dependent random 64-B lines/s, DRAM-resident set
except (OSError, KeyError, ValueError, IndexError):
    pass
# spot check against the construction
for i in rng.integers(0, args.queries, size=2000):
    r = int(rows[i]); got = [int(x) for x in labels[int(begin[i]):int(begin[i + 1])]]
    want = [j for j in (r // seg - 1, r // seg, r // seg + 1) if 0 <= j < n_labels and max(0, j * seg - 15) <= r < min(n_rows, (j + 1) * seg + 15)]
    assert got == want, (r, got, want)
print(json.dumps({"rows": n_rows, "labels": n_labels, "device_bytes": int(L.mgx_annotation_device_bytes(h)),
                  "create_s_incl_h2d_of_columns": round(t_create, 2), "queries": args.queries,
                  "get_rows_s_host_in_host_out": round(best, 4), "rows_per_s": round(args.queries / best),
                  "labels_returned": int(need.value), "spot_check": "2000 rows ok",
                  "get_rows_ms_device_resident_hip_events": [round(x, 3) for x in dev_ms],
                  "rows_per_s_device_resident": round(args.queries / dev_best),
                  "random_lines_per_s": round(lines / dev_best),
                  "frac_of_measured_random_line_ceiling": round(lines / dev_best / ceiling, 3) if ceiling else None,
                  "ceiling_lines_per_s": ceiling,
                  "note": "device-resident: rows, begin[] and labels[] in HBM, HIP events on the null stream around the call (count "
                          "kernel, scan, gather kernel); the host-in / host-out figure above includes the PCIe copies of rows and results"}))

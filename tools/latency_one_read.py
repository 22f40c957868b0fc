"""f4 (`query --align`, cli/query.cpp:1181-1190: align_sequence() builds a DBGAligner and aligns ONE sequence): latency of a
one-read batch through the C-ABI on the bench graph, host buffers in, decoded results out (mgx_align_batch).
    python tools/latency_one_read.py > profiles/rNN_latency_one_read.json"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from metagraph_amd import aligner, capi, synth  # noqa: E402

k, L = 31, 150
dev = torch.device("cuda", 0)
genome = synth.random_genome(98_000_000, 20240501, dev)
boss = synth.build_boss([genome[None, :], synth.snp_windows(genome, 200_000, k, 20240502)], k)
W, last = boss["W"].contiguous(), boss["last"].contiguous()
G = aligner.Graph(k, (W.data_ptr(), boss["n_edges"] + 1), (last.data_ptr(), boss["n_edges"] + 1), boss["F"], device=0, on_device=True)
reads = synth.sample_reads(genome, 64, L, 20240503).view(64, L).cpu().numpy()
seqs = ["".join(chr(c) for c in row) for row in reads]
A = aligner.Aligner(G, capi.config_cli(k))
out = {"graph_edges": int(boss["n_edges"]), "k": k, "read_length": L}
for n in ((1,) if os.environ.get("MGX_HOST_TIMERS") else (1, 8, 64)):
    A.align_batch(seqs[:n])                      # first call: buffers allocated
    ts = []
    for rep in range(30):
        t0 = time.perf_counter()
        res, status = A.align_batch(seqs[(rep % (64 // n)) * n:(rep % (64 // n)) * n + n])
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    out["batch_%d" % n] = {"median_ms": round(ts[len(ts) // 2], 3), "min_ms": round(ts[0], 3), "p90_ms": round(ts[int(len(ts) * 0.9)], 3)}
print(json.dumps(out))

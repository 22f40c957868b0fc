#!/usr/bin/env python3
"""BASELINE.json configs[0]: `metagraph align` of tests/data/transcripts_1000.fa against its own k = 12 DBGSuccinct graph
(the reference's CPU-runnable case).  One JSON line: sizes, GPU time of the batch (second call), the restated CPU path on
all host threads, and the number of queries whose alignments differ."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

ge.build()
import orc  # noqa: E402
from metagraph_amd import aligner, capi  # noqa: E402
from test_oracle_kats import read_fasta  # noqa: E402

orc.use_library(orc.build_fast())
seqs = read_fasta(os.path.join(ROOT, "tests", "golden", "transcripts_1000.fa"))
g = orc.Graph.build(12, seqs, 0, False)
cfg = capi.config_cli(12)
threads = os.cpu_count() or 1
t0 = time.time()
want = orc.AlignRun(g, cfg, seqs, threads=threads, validate=False).results()
t_cpu = time.time() - t0
W, last, F, valid = g.export()
G = aligner.Graph(g.k, W, last, F, valid)
A = aligner.Aligner(G, cfg)
A.align_batch(seqs)
t1 = time.time()
got, status = A.align_batch(seqs)
t_gpu = time.time() - t1
st = A.stats()
pc, xc = st["phase_cycles"], st["extend_cycles"]
tot = max(1, sum(pc[:6]))
shares = {"prepare": pc[0], "seed_pickup": pc[1], "extend": pc[2], "backtrack": pc[3], "driver": pc[4], "output": pc[5]}
ext_shares = {"pop": xc[0], "general_step": xc[1], "chain_step": xc[2]}
print(json.dumps({"config": "transcripts_1000.fa vs its own k=12 DBGSuccinct graph, CLI defaults", "queries": len(seqs),
                  "bases": sum(len(s) for s in seqs), "longest": max(len(s) for s in seqs), "graph_edges": int(g.n_edges),
                  "gpu_batch_s_host_buffers": round(t_gpu, 3),
                  "gpu_kernel_ms": {"k_map": round(st["seed_kernel_ms"], 1), "k_seed": round(st["seeding_ms"], 1), "k_extend": round(st["extend_ms"], 1)},
                  "k_extend_group_time_share": {k_: round(v / tot, 3) for k_, v in shares.items()},
                  "extend_share": {k_: round(v / tot, 3) for k_, v in ext_shares.items()},
                  "seeds_per_query": round(st["n_seeds"] / max(1, len(seqs)), 1), "extensions_per_query": round(st["n_extensions"] / max(1, len(seqs)), 1),
                  "chain_column_fraction": round(st["n_fast_columns"] / max(1, st["n_columns"]), 3),
                  "cpu_port_s": round(t_cpu, 3), "cpu_threads": threads, "columns": int(st["n_columns"]),
                  "mismatching_queries": sum(1 for a, b in zip(got, want) if a != b), "capacity_errors": sum(1 for s in status if s != 0)}), flush=True)
os._exit(0)

"""Experiment: how well does predicted_work() (first seed of the chosen strand) predict the measured
(n_extensions, n_columns) of a read?  Not part of the product or the bench."""
import collections
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from metagraph_amd import aligner, capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
k, L = 31, 150
dev = torch.device("cuda", 0)
genome = synth.random_genome(98_000_000, 20240501, dev)
boss = synth.build_boss([genome[None, :], synth.snp_windows(genome, 200_000, k, 20240502)], k)
W, last = boss["W"].contiguous(), boss["last"].contiguous()
G = aligner.Graph(k, (W.data_ptr(), boss["n_edges"] + 1), (last.data_ptr(), boss["n_edges"] + 1), boss["F"], device=0, on_device=True)
reads = synth.sample_reads(genome, n, L, 20240503).contiguous()
offsets = (torch.arange(n + 1, device=dev, dtype=torch.int64) * L).contiguous()
A = aligner.Aligner(G, capi.config_cli(k))
A.keep_seeds(True)
A.align_device(reads.data_ptr(), offsets.data_ptr(), n)
torch.cuda.synchronize()
info = A.seed_info(n)
conf = collections.Counter()
err = []
by_nseeds = collections.Counter()
for r in info:
    nm = r["num_matches"]
    first = 0 if nm[0] >= nm[1] else 1
    ss = r["seeds"][first]
    if not ss:
        pe, pc = 0, 0
    else:
        clip, length = ss[0][0], ss[0][1]
        pe = (1 if clip > 0 else 0) + (1 if clip + length < L else 0)
        pc = L - length
    conf[(pe, min(r["n_extensions"], 4))] += 1
    err.append(r["n_columns"] - pc)
    by_nseeds[(len(ss), min(r["n_extensions"], 4))] += 1
print("confusion (predicted n_ext, actual n_ext):", sorted(conf.items()))
err = np.array(err)
print("cols error: mean %.1f, |err| mean %.1f, p50 %.0f p90 %.0f p99 %.0f" % (err.mean(), np.abs(err).mean(), *np.percentile(err, [50, 90, 99])))
print("(n_seeds chosen strand, actual n_ext):", sorted(by_nseeds.items()))

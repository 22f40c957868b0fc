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
        pe = 1 + (1 if clip > 0 else 0)
        pc = (L - clip) + (L if clip > 0 else 0)
    conf[(pe, min(r["n_extensions"], 4))] += 1
    err.append(r["n_columns"] - pc)
    by_nseeds[(len(ss), min(r["n_extensions"], 4))] += 1
print("confusion (predicted n_ext, actual n_ext):", sorted(conf.items()))
err = np.array(err)
print("cols error: mean %.1f, |err| mean %.1f, p50 %.0f p90 %.0f p99 %.0f" % (err.mean(), np.abs(err).mean(), *np.percentile(err, [50, 90, 99])))
rel = np.abs(err) / np.maximum(1, np.array([r["n_columns"] for r in info]))
print("relative |err|: mean %.3f p50 %.3f p90 %.3f p99 %.3f" % (rel.mean(), *np.percentile(rel, [50, 90, 99])))
big = [(r["n_columns"], e, r["n_extensions"], len(r["seeds"][0]), len(r["seeds"][1]), r["num_matches"]) for r, e in zip(info, err) if abs(e) > 60][:12]
print("examples |err| > 60 (cols, err, n_ext, n_seeds fwd/rc, num_matches):", big)

# lock-step idle estimate: wavefronts take 8 consecutive reads of the sorted order; every extension phase lasts as
# long as its longest member
keys = []
for r in info:
    nm = r["num_matches"]
    first = 0 if nm[0] >= nm[1] else 1
    ss = r["seeds"][first]
    if not ss:
        keys.append(0)
    else:
        clip = ss[0][0]
        keys.append(1 + min(4094, (L - clip) + (L if clip > 0 else 0)))
order = np.argsort(np.array(keys), kind="stable")
cols = np.array([r["n_columns"] for r in info])[order]
m = (len(cols) // 8) * 8
b = cols[:m].reshape(-1, 8)
print("idle estimate with the predicted key: %.3f; natural order: %.3f; perfect (sorted by cols): %.3f" % (
    1 - b.mean(axis=1).sum() / b.max(axis=1).sum(),
    1 - np.array([r["n_columns"] for r in info])[:m].reshape(-1, 8).mean(axis=1).sum() / np.array([r["n_columns"] for r in info])[:m].reshape(-1, 8).max(axis=1).sum(),
    1 - np.sort(cols)[:m].reshape(-1, 8).mean(axis=1).sum() / np.sort(cols)[:m].reshape(-1, 8).max(axis=1).sum()))

nse = []
for r in info:
    nm = r["num_matches"]
    first = 0 if nm[0] >= nm[1] else 1
    nse.append(len(r["seeds"][first]))
nse = np.array(nse)
allcols = np.array([r["n_columns"] for r in info])
for thr in (6, 8, 10, 12, 16):
    k2 = np.array(keys) + np.where(nse > thr, 4096, 0)
    o2 = np.argsort(k2, kind="stable")
    c2 = allcols[o2][:m].reshape(-1, 8)
    print("n_seeds > %d flagged: %.1f%% of reads, idle %.3f" % (thr, 100.0 * (nse > thr).mean(), 1 - c2.mean(axis=1).sum() / c2.max(axis=1).sum()))
ext = np.array([r["n_extensions"] for r in info])
print("actual n_ext >= 3 share %.3f; among flagged(>8): %.3f; recall of flag(>8) on n_ext>=3: %.3f" % ((ext >= 3).mean(), (ext[nse > 8] >= 3).mean(), (nse[ext >= 3] > 8).mean()))

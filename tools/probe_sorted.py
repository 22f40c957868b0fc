"""Experiment: how much does ordering reads by their (measured) extension work help the sub-wave-group
kernels?  Runs the bench workload once, sorts the reads by (n_extensions, n_columns) of that run, and times
the aligner on the natural and on the sorted order.  Not part of the product or the bench."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from metagraph_amd import aligner, capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k, L = 31, 150
dev = torch.device("cuda", 0)
genome = synth.random_genome(98_000_000, 20240501, dev)
boss = synth.build_boss([genome[None, :], synth.snp_windows(genome, 200_000, k, 20240502)], k)
W, last = boss["W"].contiguous(), boss["last"].contiguous()
G = aligner.Graph(k, (W.data_ptr(), boss["n_edges"] + 1), (last.data_ptr(), boss["n_edges"] + 1), boss["F"], device=0, on_device=True)
reads = synth.sample_reads(genome, n, L, 20240503).contiguous()
offsets = (torch.arange(n + 1, device=dev, dtype=torch.int64) * L).contiguous()
del genome
torch.cuda.empty_cache()
A = aligner.Aligner(G, capi.config_cli(k))


def run(r, tag):
    A.align_device(r.data_ptr(), offsets.data_ptr(), n)
    torch.cuda.synchronize()
    A.align_device(r.data_ptr(), offsets.data_ptr(), n)
    torch.cuda.synchronize()
    st = A.stats()
    print(tag, {"k_map": round(st["seed_kernel_ms"], 1), "k_seed": round(st["seeding_ms"], 1), "k_extend": round(st["extend_ms"], 1)}, flush=True)


run(reads, "natural")
lib = capi.lib()
lib.mgx_fetch_seed_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
info = np.zeros(6 * n, dtype=np.uint32)
ms = C.c_uint32()
assert lib.mgx_fetch_seed_info(A.h, info.ctypes.data_as(C.POINTER(C.c_uint32)), None, C.byref(ms)) == 0
info = info.reshape(n, 6)
ext, cols = info[:, 4].astype(np.int64), info[:, 5].astype(np.int64)
print("extensions histogram", np.bincount(ext)[:8], "columns mean", cols.mean(), "p50/p90/p99", np.percentile(cols, [50, 90, 99]))
order = np.lexsort((cols, ext))
r2 = reads.view(n, L)[torch.as_tensor(order, device=dev)].contiguous().view(-1)
run(r2, "sorted by (n_ext, n_cols)")
nseed = (info[:, 2] + info[:, 3]).astype(np.int64)
order = np.lexsort((nseed, ext))
r3 = reads.view(n, L)[torch.as_tensor(order, device=dev)].contiguous().view(-1)
run(r3, "sorted by (n_ext, n_seeds)")

"""Experiment: do the mapping / seeding kernels of one half-batch overlap with the extension kernel of the other?  Two aligner
handles over one graph, two host threads, each aligning its half of the reads; the library variant is built with
-fgpu-default-stream=per-thread so that every thread's launches go to its own stream (MGX_LIB_PATH selects it).  Prints the wall time of
(a) one handle aligning all reads and (b) two threads aligning the halves concurrently, `stagger` seconds apart."""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metagraph_amd import aligner, capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
stagger = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
k, L = 31, 150
dev = torch.device("cuda", 0)
genome = synth.random_genome(98_000_000, 20240501, dev)
boss = synth.build_boss([genome[None, :], synth.snp_windows(genome, 200_000, k, 20240502)], k)
W, last = boss["W"].contiguous(), boss["last"].contiguous()
G = aligner.Graph(k, (W.data_ptr(), boss["n_edges"] + 1), (last.data_ptr(), boss["n_edges"] + 1), boss["F"], device=0, on_device=True)
reads = synth.sample_reads(genome, n, L, 20240503).contiguous()
h = n // 2
offs_all = (torch.arange(n + 1, device=dev, dtype=torch.int64) * L).contiguous()
offs_half = (torch.arange(h + 1, device=dev, dtype=torch.int64) * L).contiguous()
del genome
torch.cuda.empty_cache()
cfg = capi.config_cli(k)
A = [aligner.Aligner(G, cfg), aligner.Aligner(G, cfg)]
flat = reads.view(-1)
halves = [flat[:h * L].contiguous(), flat[h * L:].contiguous()]


def whole():
    A[0].align_device(reads.data_ptr(), offs_all.data_ptr(), n)
    torch.cuda.synchronize()


def two():
    def work(i):
        if i:
            time.sleep(stagger)
        try:
            A[i].align_device(halves[i].data_ptr(), offs_half.data_ptr(), h)
        except Exception as e:
            print("thread", i, "failed:", e, "handle", A[i].h, "seqs %x offsets %x" % (halves[i].data_ptr(), offs_half.data_ptr()), flush=True)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()


for name, fn in (("whole batch, one handle", whole), ("two halves, two threads", two), ("whole batch, one handle", whole), ("two halves, two threads", two)):
    fn()                                   # warm-up (arena allocation)
    t0 = time.time()
    fn()
    dt = time.time() - t0
    st = [a.stats() for a in A]
    print("%-28s %.3f s  (%.2f M reads/s)  k_extend per handle: %s ms" % (name, dt, n / dt / 1e6, [round(s["extend_ms"], 1) for s in st]), flush=True)

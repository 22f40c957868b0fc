"""The bench fixture builder (metagraph_amd/synth.py, torch) against the oracle's BOSS fixture builder."""
import numpy as np
import pytest
import torch

import orc
from metagraph_amd import synth


@pytest.mark.parametrize("k,n,snps,seed", [(9, 3000, 30, 1), (5, 400, 10, 2), (31, 5000, 40, 3), (12, 2000, 0, 4)])
def test_boss_builder_matches_oracle(k, n, snps, seed):
    dev = torch.device("cpu")
    genome = synth.random_genome(n, seed, dev)
    tensors = [genome[None, :]]
    seqs = ["".join(synth.CHARS[c] for c in genome.tolist())]
    if snps:
        win = synth.snp_windows(genome, snps, k, seed + 100)
        tensors.append(win)
        seqs += ["".join(synth.CHARS[c] for c in row) for row in win.tolist()]
    b = synth.build_boss(tensors, k)
    g = orc.Graph.build(k, seqs, 0, False)
    W, last, F, _ = g.export()
    assert b["n_edges"] == g.n_edges
    assert list(F) == b["F"]
    assert np.array_equal(b["last"].numpy(), last)
    assert np.array_equal(b["W"].numpy(), W)


def test_reads_shape_and_alphabet():
    dev = torch.device("cpu")
    genome = synth.random_genome(5000, 9, dev)
    r = synth.sample_reads(genome, 200, 150, 5)
    assert r.shape == (200, 150)
    assert set(np.unique(r.numpy()).tolist()) <= {ord(c) for c in "ACGT"}


def test_bench_workload_declared_primary():
    """`bench.py --graph-mode primary` declares the synthetic BOSS table PRIMARY: the forward strand of an iid genome (plus SNP
    windows) holds one k-mer of every {k-mer, reverse complement} pair, so it is a valid primary graph.  The bench's own
    pieces at small scale on the CPU: generator -> BOSS table -> oracle (mgx_boss_view.mode = PRIMARY, as the bench's CPU leg
    builds it) and the kernels' host model, same reads."""
    import ctypes as C
    import emu_drv
    from metagraph_amd import capi
    dev = torch.device("cpu")
    k = 31
    genome = synth.random_genome(60000, 20240501, dev)
    b = synth.build_boss([genome[None, :], synth.snp_windows(genome, 120, k, 20240502)], k)
    W, last = b["W"].numpy().copy(), b["last"].numpy().copy()
    view, keep = emu_drv.boss_view(k, W, last, b["F"], None, mode=2)
    og = orc.Graph(orc.L().orc_graph_from_boss(C.byref(view)))
    reads = [bytes(r).decode() for r in synth.sample_reads(genome, 300, 150, 20240503).numpy()]
    cfg = capi.config_cli(k)
    want = orc.AlignRun(og, cfg, reads)
    assert want.error == ""
    eg = emu_drv.EmuGraph.__new__(emu_drv.EmuGraph)
    eg.view, eg._keep, eg.k = view, keep, k
    eg.h = emu_drv.L().emu_graph_create(C.byref(view))
    got, status = emu_drv.EmuRun(eg, cfg, reads).results()
    assert all(s == 0 for s in status) and got == want.results()
    aligned = sum(1 for q in got if q)
    assert aligned > 0.85 * len(reads)
    n = b["n_edges"]
    assert any(v > n for q in got for a in q for v in a["nodes"])          # reverse-strand reads resolve to ids above n

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

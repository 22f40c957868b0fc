"""Size-independent properties of the HIP path at the bench workload's graph size (no oracle: it could not finish these sizes):
the results of a read depend on nothing but the read — not on its position in the batch (work sort, arena slot, wavefront
mates, multi-pass schedule all change under a permutation), not on what the aligner handle ran before (idempotence).  Per-query
digests over every field of every alignment (tests/result_digest.py, validated on the oracle in tests/test_result_digest.py).
Default: the 98 Mbp bench graph (104 M edges, k = 31) with 1 M reads (the digests take ~5 GB of host memory per million reads);
MGX_PROP_READS=10000000 runs BASELINE's full batch."""
import os

import numpy as np
import pytest

from result_digest import query_digests

pytestmark = pytest.mark.gpu


def test_results_do_not_depend_on_batch_order_or_history():
    import torch
    from metagraph_amd import aligner, capi, synth
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    k, read_len = 31, 150
    genome_len = int(os.environ.get("MGX_PROP_GENOME", 98_000_000))
    n_reads = int(os.environ.get("MGX_PROP_READS", 1_000_000))
    genome = synth.random_genome(genome_len, 20240501, dev)
    boss = synth.build_boss([genome[None, :], synth.snp_windows(genome, genome_len // 490, k, 20240502)], k)
    torch.cuda.synchronize()
    n_edges = boss["n_edges"]
    W, last = boss["W"].contiguous(), boss["last"].contiguous()
    G = aligner.Graph(k, (W.data_ptr(), n_edges + 1), (last.data_ptr(), n_edges + 1), boss["F"], device=0, on_device=True)
    reads = synth.sample_reads(genome, n_reads, read_len, 777).contiguous()
    offsets = (torch.arange(n_reads + 1, device=dev, dtype=torch.int64) * read_len).contiguous()
    del genome, boss, W, last
    torch.cuda.empty_cache()
    A = aligner.Aligner(G, capi.config_cli(k))

    def run(batch):
        torch.cuda.synchronize()
        A.align_device(batch.data_ptr(), offsets.data_ptr(), n_reads)
        res = A.fetch()
        assert res.n_queries == n_reads
        d = query_digests(res)                        # (the views die with the next batch: digest now)
        aligned = int(np.count_nonzero(np.diff(capi.results_arrays(res)["aln_begin"].astype(np.int64))))
        return d, aligned

    d1, aligned = run(reads)
    assert aligned > 0.85 * n_reads                   # ~5 % of the synthetic reads are random sequence
    perm = torch.randperm(n_reads, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    shuffled = reads[perm].contiguous()
    d2, aligned2 = run(shuffled)
    assert aligned2 == aligned
    assert np.array_equal(d2, d1[perm.cpu().numpy()])          # permuting the batch permutes the results, nothing else
    d3, _ = run(reads)
    assert np.array_equal(d3, d1)                              # and the handle's history leaves no trace

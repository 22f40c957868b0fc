"""The lane-per-read kernel (mgx_lane.hip) on the GPU: forced in front of the group kernel (option lane=1) on worlds of the
benchmark's shape and on the parity suite's, every read against the oracle; and the automatic choice on a batch large enough
to take it.  (tests/test_lane_read.py runs the same per-read function in the host model.)"""
import os
import random

import pytest

import orc
from metagraph_amd import aligner, capi
from test_emu_vs_oracle import make_world, mutate
from test_gpu_parity import compare_gpu, gpu_graph
from test_lane_read import bench_like_world

pytestmark = pytest.mark.gpu


def test_lane_kernel_on_bench_like_reads():
    g, reads = bench_like_world(11, 20000, genome_len=150000)
    cfg = capi.config_cli(31)
    A = aligner.Aligner(gpu_graph(g), cfg)
    A.set_pipeline("lane=1")
    got, status = A.align_batch(reads)
    assert all(s == 0 for s in status)
    st = A.stats()
    assert st["extend_kernels"] & capi.KERNEL_LANE and st["n_lane_reads"] > 0.7 * len(reads), st
    want = orc.AlignRun(g, cfg, reads, threads=os.cpu_count() or 8, validate=False).results()
    for q in range(len(reads)):
        assert got[q] == want[q], (q, reads[q], got[q], want[q])
    # the same batch again through the same handle (node tables of the first launch must read as empty), then without the kernel
    got2, _ = A.align_batch(reads)
    assert got2 == want and A.stats()["n_lane_reads"] == st["n_lane_reads"]
    A.set_pipeline("lane=0")
    got3, _ = A.align_batch(reads)
    assert got3 == want and A.stats()["n_lane_reads"] == 0 and not (A.stats()["extend_kernels"] & capi.KERNEL_LANE)


@pytest.mark.parametrize("snp_every", [490, 120])
def test_lane_kernel_on_reads_over_snp_bubbles(snp_every):
    """the benchmark's graph shape: bubbles the extensions fork at — behind the query's end the children of a fork tie level
    by level (LANE_TIE_MODE), a later seed may end in the first seed's first node (the merged vector of the replay columns)"""
    g, reads = bench_like_world(20 + snp_every, 20000, genome_len=120000, snp_every=snp_every)
    cfg = capi.config_cli(31)
    A = aligner.Aligner(gpu_graph(g), cfg)
    A.set_pipeline("lane=1")
    got, status = A.align_batch(reads)
    assert all(s == 0 for s in status)
    st = A.stats()
    assert st["extend_kernels"] & capi.KERNEL_LANE and st["n_lane_reads"] > (0.9 if snp_every == 490 else 0.7) * len(reads), st
    want = orc.AlignRun(g, cfg, reads, threads=os.cpu_count() or 8, validate=False).results()
    for q in range(len(reads)):
        assert got[q] == want[q], (q, reads[q], got[q], want[q])


@pytest.mark.parametrize("seed", range(4))
def test_lane_kernel_on_random_worlds(seed):
    rng = random.Random(2000 + seed)
    k = rng.choice([11, 15, 21, 31])
    g, reads = make_world(2000 + seed, k, genome_len=9000, n_reads=900, read_len=rng.choice([100, 150, 230]), n_variants=rng.choice([0, 40]),
                          mask=seed == 3)
    reads = [mutate(rng, x, sub=0.03, ins=0.01, dele=0.01) if i % 3 == 0 else x for i, x in enumerate(reads)]
    reads += ["", "ACGT", "N" * 80, "A" * 90, reads[0][:40]]
    cfg = capi.config_cli(k)
    if seed % 2:
        cfg.left_end_bonus, cfg.right_end_bonus = 2, 3
    # (validate=False: the reference's own Alignment::is_valid — a debug assertion there — rejects some alignments it produces
    # with end bonuses; the comparison is against what it produces)
    want = orc.AlignRun(g, cfg, reads, validate=False).results()
    A = aligner.Aligner(gpu_graph(g), cfg)
    for opt in ("lane=1", "ext64=0"):
        A.set_pipeline(opt)
    got, status = A.align_batch(reads)
    assert all(s == 0 for s in status)
    for q in range(len(reads)):
        assert got[q] == want[q], (q, reads[q], got[q], want[q])
    st = A.stats()
    assert st["extend_kernels"] & capi.KERNEL_LANE and st["n_lane_reads"] > 0


def test_lane_kernel_is_the_automatic_choice_for_large_batches_only():
    g, reads = bench_like_world(12, 40000, genome_len=100000)
    cfg = capi.config_cli(31)
    G = gpu_graph(g)
    A = aligner.Aligner(G, cfg)
    got, status = A.align_batch(reads)
    assert all(s == 0 for s in status) and A.stats()["extend_kernels"] & capi.KERNEL_LANE
    want = orc.AlignRun(g, cfg, reads, threads=os.cpu_count() or 8, validate=False).results()
    assert got == want
    got, status = A.align_batch(reads[:500])
    assert not (A.stats()["extend_kernels"] & capi.KERNEL_LANE) and got == want[:500]

"""The lane-per-read seeder (mgx_seedlane.hip, seed_lane.hpp) on the GPU: forced in front of the seeding kernel (option
seed_lane=1), seed lists, num_matching and alignments of every read against the oracle; the reads it leaves are seeded by the
wave program in the same batch.  (tools/fuzz_emu.py --seedlane and tests/test_seed_lane.py run the same per-read function in
the host model.)"""
import os
import random

import pytest

import orc
from emu_drv import oracle_seeds_as_tuples
from metagraph_amd import aligner, capi
from test_emu_vs_oracle import make_world, mutate
from test_gpu_parity import gpu_graph
from test_lane_read import bench_like_world

pytestmark = pytest.mark.gpu


def check(g, cfg, reads, options, min_share):
    o = orc.AlignRun(g, cfg, reads, threads=os.cpu_count() or 8, validate=False)
    assert o.error == "", o.error
    A = aligner.Aligner(gpu_graph(g), cfg)
    for opt in options:
        A.set_pipeline(opt)
    A.keep_seeds(True)
    got, status = A.align_batch(reads)
    assert all(s == 0 for s in status)
    st = A.stats()
    if min_share is not None:
        assert st["n_seed_lane_reads"] >= min_share * len(reads), (st["n_seed_lane_reads"], st["seed_lane_left_reads"])
    info = A.seed_info(len(reads))
    for strand in (0, 1):
        for q, (ss, nm) in enumerate(o.seeds(strand)):
            assert info[q]["num_matches"][strand] == nm, (q, strand, reads[q])
            assert info[q]["seeds"][strand] == oracle_seeds_as_tuples(ss), (q, strand, reads[q])
    want = o.results()
    for q in range(len(reads)):
        assert got[q] == want[q], (q, reads[q], got[q], want[q])
    return A, st


@pytest.mark.parametrize("snp_every", [0, 120])
def test_seed_lane_on_bench_like_reads(snp_every):
    g, reads = bench_like_world(31 + snp_every, 12000, genome_len=150000, **({"snp_every": snp_every} if snp_every else {}))
    reads += ["", "ACGT", "N" * 80, "A" * 150, reads[0][:40], reads[1][:31], reads[2][:10] + "N" + reads[2][11:]]
    cfg = capi.config_cli(31)
    A, st = check(g, cfg, reads, ("seed_lane=1",), 0.7)
    # the same batch through the same handle again, then without the kernel: the same answers
    got, _ = A.align_batch(reads)
    A.set_pipeline("seed_lane=0")
    got0, _ = A.align_batch(reads)
    assert got0 == got and A.stats()["n_seed_lane_reads"] == 0


@pytest.mark.parametrize("seed", range(6))
def test_seed_lane_on_random_worlds(seed):
    rng = random.Random(4100 + seed)
    k = rng.choice([11, 15, 21, 31])
    g, reads = make_world(4100 + seed, k, genome_len=9000, n_reads=900, read_len=rng.choice([60, 100, 150]), n_variants=rng.choice([0, 40]),
                          mask=seed == 3)
    reads = [mutate(rng, x, sub=0.03, ins=0.01, dele=0.01) if i % 3 == 0 else x for i, x in enumerate(reads)]
    reads += ["", "ACGT", "N" * 80, "A" * 90, reads[0][:40], "ACACACACACACACACACACACACACACACACACACACACACACACACAC"]
    cfg = capi.config_cli(k)
    if seed % 2:
        cfg.min_seed_length = rng.randrange(max(3, k // 3), k + 1)
        cfg.max_num_seeds_per_locus = rng.choice([1, 2, 1000])
    if seed == 5:
        cfg.min_exact_match = 0.7
    check(g, cfg, reads, ("seed_lane=1", "lane=1"), None)


def test_seed_lane_on_reads_of_250_characters():
    """the build of the kernel for batches with reads of 161 .. 255 characters (nine packed words per strand)"""
    g, reads = bench_like_world(250, 6000, genome_len=120000, read_len=250, snp_every=300)
    reads += [reads[0][:255], reads[1][:160], reads[2][:161], reads[3] + reads[4][:6], "ACGT" * 60]
    cfg = capi.config_cli(31)
    check(g, cfg, reads, ("seed_lane=1",), 0.9)
    cfg.max_seed_length = 31
    check(g, cfg, reads[:2000], ("seed_lane=1",), 0.8)


def test_seed_lane_is_the_automatic_choice_for_large_batches_only():
    g, reads = bench_like_world(77, 6000, genome_len=60000)
    cfg = capi.config_cli(31)
    A = aligner.Aligner(gpu_graph(g), cfg)
    A.align_batch(reads[:500])
    assert A.stats()["n_seed_lane_reads"] == 0
    A.align_batch(reads)
    assert A.stats()["n_seed_lane_reads"] > 0.7 * len(reads)

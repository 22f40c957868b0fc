"""The lane-per-read seeder (metagraph_amd/csrc/seed_lane.hpp) in the host model: the split pipeline with MGX_EMU_SEEDLANE=1 runs
seed_lane_read() on every read — two passes, as mgx.hip launches them — and the wave program's seeder on what they leave; seed
lists, num_matching and alignments of every read against the oracle.  (tools/fuzz_emu.py --seedlane: the campaign;
tests/test_gpu_seed_lane.py: the kernel on the GPU.)"""
import random

import pytest

import emu_drv
import orc
from metagraph_amd import capi
from test_emu_vs_oracle import make_world, mutate
from test_lane_read import bench_like_world


@pytest.fixture
def seedlane_env(monkeypatch):
    monkeypatch.setenv("MGX_EMU_SPLIT", "1")
    monkeypatch.setenv("MGX_EMU_SEEDLANE", "1")


def run(g, cfg, reads, mode=0):
    o = orc.AlignRun(g, cfg, reads, validate=False)
    assert o.error == "", o.error
    e = emu_drv.EmuRun(emu_drv.EmuGraph(g, mode=mode), cfg, reads)
    assert e.error == "", e.error
    got, status = e.results()
    assert all(s == 0 for s in status)
    info = e.seed_info()
    for strand in (0, 1):
        for q, (ss, nm) in enumerate(o.seeds(strand)):
            assert info[q]["num_matches"][strand] == nm, (q, strand, reads[q])
            assert info[q]["seeds"][strand] == emu_drv.oracle_seeds_as_tuples(ss), (q, strand, reads[q])
    assert got == o.results()
    return e.seedlane_stats()


@pytest.mark.parametrize("snp_every", [0, 120])
def test_bench_like_reads(seedlane_env, snp_every):
    g, reads = bench_like_world(41 + snp_every, 1500, genome_len=60000, snp_every=snp_every)
    reads += ["", "ACGT", "N" * 80, "A" * 150, reads[0][:40], reads[1][:31], reads[2][:10] + "N" + reads[2][11:],
              "ACAC" * 37, "ACGTTGCA" * 18]
    ran, done, why = run(g, capi.config_cli(31), reads)
    assert ran and done > 0.9 * len(reads), (done, why)
    assert set(why) <= {1, 2, 3, 4, 9, 11}, why               # (a read length, invalid characters, DUST, a look-up that disagrees)


def test_first_pass_alone(seedlane_env, monkeypatch):
    monkeypatch.setenv("MGX_EMU_SEEDLANE_ONE", "1")
    g, reads = bench_like_world(43, 800, genome_len=40000, snp_every=200)
    ran, done, why = run(g, capi.config_cli(31), reads)
    assert ran and 0.7 * len(reads) < done and 8 in why, (done, why)      # (strands below min_exact_match wait for the second pass)


@pytest.mark.parametrize("seed", range(6))
def test_random_worlds(seedlane_env, seed):
    rng = random.Random(5200 + seed)
    k = rng.choice([7, 11, 15, 21, 31])
    g, reads = make_world(5200 + seed, k, genome_len=5000, n_reads=200, read_len=min(rng.choice([60, 100, 150]), k + 110), n_variants=rng.choice([0, 40]),
                          mask=seed == 3)
    reads = [mutate(rng, x, sub=0.03, ins=0.01, dele=0.01) if i % 3 == 0 else x for i, x in enumerate(reads)]
    cfg = capi.config_cli(k)
    if seed % 2:
        cfg.min_seed_length = rng.randrange(max(3, k // 3), k + 1)
        cfg.max_num_seeds_per_locus = rng.choice([1, 2, 1000])
    if seed == 5:
        cfg.min_exact_match = 0.0
    if seed == 4:
        cfg.seed_complexity_filter = 0
    ran, done, why = run(g, cfg, reads)
    assert ran and done > 0, why


def test_one_seed_per_kmer(seedlane_env):
    """max_seed_length == k (what label-aware alignment sets): ExactSeeder's seed per matched k-mer, every tail position looked up"""
    g, reads = bench_like_world(47, 600, genome_len=40000, snp_every=150)
    cfg = capi.config_cli(31)
    cfg.max_seed_length = 31
    ran, done, why = run(g, cfg, reads)
    assert ran and done > 0.9 * len(reads), (done, why)


@pytest.mark.parametrize("many", [0, 1])
def test_low_complexity_islands(seedlane_env, many):
    """reads over homopolymer / short tandem stretches: the second pass answers the DUST filter's windows from its map of the
    read (seed_lane.hpp sl_dust_map) — masked sub-k positions report nothing, masked k-mer seeds are dropped"""
    rng = random.Random(91 + many)
    from test_emu_vs_oracle import rand_seq, rc
    G = list(rand_seq(rng, 30000))
    for _ in range(120):
        a = rng.randrange(0, len(G) - 60)
        u = rng.choice(["A", "C", "T", "AT", "CG", "AC", "AAT", "CAG", "ACGT"])
        for x in range(a, a + rng.randint(6, 45)):
            G[x] = u[(x - a) % len(u)]
    G = "".join(G)
    k = 21
    g = orc.Graph.build(k, [G], 0, False)
    reads = []
    for _ in range(700):
        p = rng.randrange(0, len(G) - 130)
        r = mutate(rng, G[p:p + 120], sub=0.02, ins=0.002, dele=0.002)
        reads.append(rc(r) if rng.random() < 0.5 else r)
    cfg = capi.config_cli(k)
    cfg.min_seed_length = 12
    if many:
        cfg.max_seed_length = k
    ran, done, why = run(g, cfg, reads)
    assert ran and done > 0.85 * len(reads), (done, why)


def test_configurations_the_kernel_does_not_take(seedlane_env):
    g, reads = make_world(77, 15, genome_len=3000, n_reads=40, read_len=80)
    cfg = capi.config_cli(15)
    cfg.max_seed_length = 10                                   # below k: the k-mers are not even mapped
    cfg.min_seed_length = 8
    ran, done, why = run(g, cfg, reads)
    assert not ran and done == 0

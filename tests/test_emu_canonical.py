"""CANONICAL-mode graphs through the kernels (host model) against the oracle: seeds, num_matches and full alignment lists on
the reference's canonical KATs, the genome.MT canonical CLI goldens and seeded random worlds; also the split pipeline, 8-lane
groups and the multi-pass extension."""
import os
import random

import pytest

import emu_drv
import orc
from metagraph_amd import capi
from test_emu_vs_oracle import compare_full, rand_seq, mutate, rc
from test_oracle_kats import read_fasta, read_fastq, HERE
from test_oracle_canonical import CANONICAL, CANONICAL_LINES, SUBK_LINE_5, _cfg


def canonical_world(seed, k, genome_len=3000, n_reads=40, read_len=100, mask=False, n_variants=10):
    rng = random.Random(seed)
    genome = rand_seq(rng, genome_len)
    seqs = [genome]
    for _ in range(n_variants):
        p = rng.randrange(k, genome_len - k)
        alt = rng.choice([c for c in "ACGT" if c != genome[p]])
        seqs.append(genome[p - k + 1:p] + alt + genome[p + 1:p + k])
    g = orc.Graph.build(k, seqs, CANONICAL, mask)
    reads = []
    for i in range(n_reads):
        if i % 10 == 9:
            reads.append(rand_seq(rng, read_len))
            continue
        p = rng.randrange(0, genome_len - read_len)
        r = mutate(rng, genome[p:p + read_len])
        if rng.random() < 0.5:
            r = rc(r)
        if i % 7 == 3:
            r = r[:len(r) // 2] + "N" + r[len(r) // 2 + 1:]
        reads.append(r)
    return g, reads


def test_canonical_kats_through_the_kernels():
    g = orc.Graph.build(7, ["AAAAGCTTTCGAGGCCAA"], CANONICAL, True)
    compare_full(g, emu_drv.EmuGraph(g, mode=CANONICAL), _cfg(), ["AAAAGTTTTCGAGGCCAA"])
    g = orc.Graph.build(18, ["TTGGCCTCGAAAGTTTTT"], CANONICAL, False)
    cfg = _cfg(max_num_seeds_per_locus=capi.UINT64_MAX, min_cell_score=-2147483648 + 100, min_path_score=-2147483648 + 100,
               min_seed_length=13)
    compare_full(g, emu_drv.EmuGraph(g, mode=CANONICAL), cfg, ["GGGGGCTTTCGAGGCCAA"])


@pytest.mark.parametrize("min_seed_length", [None, 10])
def test_canonical_cli_goldens_through_the_kernels(min_seed_length):
    g = orc.Graph.build(11, read_fasta(os.path.join(HERE, "golden", "genome.MT.fa")), CANONICAL, True)
    reads = read_fastq(os.path.join(HERE, "golden", "genome_MT1.fq"))
    cfg = capi.config_cli(11)
    cfg.min_exact_match = 0.0
    if min_seed_length is not None:
        cfg.min_seed_length = min_seed_length
    e = compare_full(g, emu_drv.EmuGraph(g, mode=CANONICAL), cfg, [r[1] for r in reads])
    got, _ = e.results()
    assert all(a["orientation"] == 0 for q in got for a in q)          # canonical graphs report forward alignments only
    assert (got[3][0]["cigar"], got[3][0]["score"]) == ("54=1X95=", 305)
    assert CANONICAL_LINES and SUBK_LINE_5


@pytest.mark.parametrize("k,mask,seed", [(11, False, 1), (19, False, 2), (31, False, 3), (15, True, 4)])
def test_canonical_random_worlds(k, mask, seed):
    g, reads = canonical_world(900 + seed, k, mask=mask)
    compare_full(g, emu_drv.EmuGraph(g, mode=CANONICAL), capi.config_cli(k), reads)


def test_canonical_split_multipass_and_alternatives(monkeypatch):
    monkeypatch.setenv("MGX_EMU_SPLIT", "1")
    monkeypatch.setenv("MGX_EMU_MULTIPASS", "1")
    g, reads = canonical_world(950, 15, genome_len=4000, n_reads=40, n_variants=40)
    chim = [reads[i][:50] + reads[i + 1][40:90] for i in range(0, 20, 2)]
    eg = emu_drv.EmuGraph(g, mode=CANONICAL)
    for n_alt, msl in ((1, 15), (2, 11)):
        cfg = capi.config_cli(15)
        cfg.min_exact_match = 0.0
        cfg.num_alternative_paths = n_alt
        cfg.min_seed_length = msl
        compare_full(g, eg, cfg, reads + chim)

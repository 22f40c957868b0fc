"""k_map's request / response machine (metagraph_amd/csrc/map_pipe.hpp) against the one-step-per-lane machine of rounds
1-4 (graph_build.hpp, map_lane_step_packed) and against the oracle's map_to_nodes: node arrays, match-length bytes and
(rl, ru) ranges byte for byte.  CPU only: the host model steps the very same header the HIP kernel is compiled from."""
import os
import random

import pytest

import emu_drv
import orc
from metagraph_amd import capi
from test_emu_vs_oracle import make_world, rand_seq


MLEN_TAIL = 253


def without_tails(side):
    """map_side() with the read-tail entries (MLEN_TAIL: only the pipe writes them) blanked"""
    out = []
    for mlen, rng in side:
        ml, rg = bytearray(mlen), bytearray(rng)
        for i, v in enumerate(ml):
            if v == MLEN_TAIL:
                ml[i] = 255
                rg[8 * i:8 * i + 8] = bytes(8)
        out.append((bytes(ml), bytes(rg)))
    return out


def check_tails(g, k, reads, run):
    """every MLEN_TAIL entry holds index_range of the tail position behind it: the edges whose node ends with the k - 2
    characters after the last k-mer's first two, found here by brute force over the graph's spellings"""
    import struct
    from test_emu_vs_oracle import rc
    spell = {v: g.node_sequence(v) for v in range(1, g.n_edges + 1)}
    n_checked = 0
    for strand, (mlen, rng) in enumerate(run.map_side()):
        nb = 0
        for q, r in enumerate(reads):
            nk = max(0, len(r) - k + 1)
            if nk and mlen[nb + nk - 1] == MLEN_TAIL:
                s = r.upper() if strand == 0 else rc(r.upper())
                tail = s[len(s) - (k - 2):]
                rl, ru = struct.unpack_from("<II", rng, 8 * (nb + nk - 1))
                want = [v for v, sp in spell.items() if sp[:k - 1].endswith(tail)]
                assert want and (rl, ru) == (min(want), max(want)) and len(want) == ru - rl + 1, (q, strand, rl, ru, want[:4])
                n_checked += 1
            nb += nk
    return n_checked


def both_machines(eg, cfg, reads):
    os.environ.pop("MGX_MAP_LANES", None)
    a = emu_drv.EmuRun(eg, cfg, reads, map_only=True)
    os.environ["MGX_MAP_LANES"] = "1"
    try:
        b = emu_drv.EmuRun(eg, cfg, reads, map_only=True)
    finally:
        os.environ.pop("MGX_MAP_LANES", None)
    return a, b


@pytest.mark.parametrize("k,mask,seed", [(31, False, 1), (31, True, 2), (21, False, 3), (12, False, 4), (5, False, 5), (32, False, 6),
                                         (20, True, 7), (3, False, 8)])
def test_pipe_equals_lane_machine_and_oracle(k, mask, seed):
    g, reads = make_world(100 + seed, k, genome_len=4000, n_reads=60, read_len=150, mask=mask, n_variants=25)
    rng = random.Random(seed)
    # edge cases of the chain fetch and the word top-up: reads shorter than k, of exactly k, of 32 / 33 / 64 / 65 characters,
    # runs of invalid characters across a word boundary, an empty read
    reads += ["", "A", rand_seq(rng, k - 1) if k > 1 else "", reads[0][:k], reads[1][:32], reads[2][:33], reads[3][:64], reads[4][:65],
              reads[5][:20] + "N" * 40 + reads[5][60:], "N" * 70, reads[6][:96], reads[7][:97] + "n"]
    eg = emu_drv.EmuGraph(g)
    cfg = capi.config_cli(k)
    for msl in (None, 3):
        if msl is not None:
            cfg.min_seed_length = min(msl, k)
        a, b = both_machines(eg, cfg, reads)
        assert a.mapping() == b.mapping()
        assert without_tails(a.map_side()) == b.map_side()
        assert a.mapping() == orc.AlignRun(g, cfg, reads).mapping()
        n_tails = check_tails(g, k, reads, a)
        assert n_tails > 0 or k < 8 or cfg.min_seed_length > k - 2


def test_pipe_on_a_repetitive_graph_with_wide_ranges():
    # low-complexity genome: wide suffix ranges (rl - 1 and ru in different blocks, r_lo outside the block of r_hi)
    rng = random.Random(9)
    unit = rand_seq(rng, 7)
    genome = "".join(unit if rng.random() < 0.8 else rand_seq(rng, 7) for _ in range(700))
    k = 15
    g = orc.Graph.build(k, [genome], 0, False)
    reads = [genome[p:p + 120] for p in range(0, 3000, 37)] + [rand_seq(rng, 100) for _ in range(10)]
    eg = emu_drv.EmuGraph(g)
    cfg = capi.config_cli(k)
    cfg.min_seed_length = 8
    a, b = both_machines(eg, cfg, reads)
    assert a.mapping() == b.mapping() == orc.AlignRun(g, cfg, reads).mapping()
    assert without_tails(a.map_side()) == b.map_side()
    assert check_tails(g, k, reads, a) > 0

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
        assert a.map_side() == b.map_side()
        assert a.mapping() == orc.AlignRun(g, cfg, reads).mapping()


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
    a, b = both_machines(eg, cfg, reads)
    assert a.mapping() == b.mapping() == orc.AlignRun(g, cfg, reads).mapping()
    assert a.map_side() == b.map_side()

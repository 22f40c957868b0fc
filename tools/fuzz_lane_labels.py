#!/usr/bin/env python3
"""Differential fuzzing of label-aware alignment on the lane-per-read path (lane_read.hpp, round 6) in the host model against the
oracle's LabeledAligner: segment-labelled worlds (tests/test_lane_labels.py::segment_world) with random k, read length, SNP
density, label count and configuration; alignments, label lists, the label filter's seed lists and num_matching of every read.
    python tools/fuzz_lane_labels.py N_WORLDS [FIRST_SEED]
CPU only (test infrastructure)."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MGX_EMU_SPLIT"] = "1"; os.environ["MGX_EMU_LANE"] = "1"
os.environ.setdefault("MGX_EMU_SEEDLANE", "1")          # (round 6: the lane-per-read seeder in front, one seed per k-mer)
import emu_drv, orc
from metagraph_amd import capi
from labeled_worlds import with_labels
from test_lane_labels import segment_world
tot = done_t = bad_t = sl_t = 0
first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
for seed in range(first, first + int(sys.argv[1])):
    rng = random.Random(seed)
    k = rng.choice([11, 15, 21, 27, 31])
    g, anno, reads = segment_world(seed, k=k, genome_len=rng.choice([3000, 8000]), n_labels=rng.choice([1, 3, 8]), n_reads=120,
                                   read_len=rng.choice([60, 100, 150]), snp_every=rng.choice([0, 90, 200]) or 10**9, label_alt=bool(seed % 2))
    cfg = capi.config_cli(k)
    if seed % 5 == 0: cfg.min_seed_length = max(5, k - 6)
    if seed % 7 == 0: cfg.min_exact_match = 0.9
    if seed % 11 == 0: cfg.xdrop = 10
    o = orc.LabeledAlignRun(g, cfg, anno, reads)
    e = emu_drv.EmuRun(emu_drv.EmuGraph(g), cfg, reads, annotation=emu_drv.EmuAnnotation(anno))
    assert e.error == "" and o.error == "", (e.error, o.error)
    got, status = e.results()
    want = with_labels(o)
    bad = [q for q in range(len(reads)) if got[q] != want[q] or status[q]]
    info = e.seed_info()
    for strand in (0, 1):
        for q, (ss, nm) in enumerate(o.seeds(strand)):
            if info[q]["num_matches"][strand] != nm or info[q]["seeds"][strand] != emu_drv.oracle_seeds_as_tuples(ss): bad.append(q)
    ran, nd = e.lane_stats()
    sl_t += e.seedlane_stats()[1]
    tot += len(reads); done_t += nd; bad_t += len(bad)
    if bad: print("seed", seed, "k", k, "BAD", bad[:5])
print("worlds done: reads", tot, "seeded by the lane-per-read seeder", sl_t, "lane finished", done_t, "mismatches", bad_t)

#!/usr/bin/env python3
"""Differential fuzzing of the lane-per-read kernels ON THE GPU (libmgx.so as built: k_seed_lane, k_lane — the unit compiled with
the iterative-ilp scheduling strategy — in front of the group kernel) against the oracle: random worlds of the parity suite's and of
the benchmark's shape, random scoring / seeding configurations, every read of every world.
    python tools/fuzz_gpu_lane.py MINUTES [FIRST_SEED] [--labels]
--labels: label-aware alignment (LabeledAligner) on segment-labelled worlds (tests/test_lane_labels.py::segment_world), alignments and
label lists against the oracle's LabeledAligner.
Needs a GPU (run through gpurun).  Test infrastructure: the oracle is the checker."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
from metagraph_amd import aligner, capi
from test_emu_vs_oracle import make_world, mutate
from test_gpu_parity import gpu_graph
from test_lane_read import bench_like_world

labels = "--labels" in sys.argv
argv = [a for a in sys.argv if a != "--labels"]
minutes = float(argv[1]) if len(argv) > 1 else 5.0
seed = first = int(argv[2]) if len(argv) > 2 else 9000
if labels:
    from labeled_worlds import with_labels
    from test_lane_labels import segment_world
    from test_gpu_labels import gpu_annotation
t0 = time.time()
worlds = reads_total = lane_total = seeded_total = 0
while time.time() - t0 < 60 * minutes:
    rng = random.Random(seed)
    if labels:
        k = rng.choice([11, 15, 21, 27, 31])
        g, anno, reads = segment_world(seed, k=k, genome_len=rng.choice([3000, 8000, 20000]), n_labels=rng.choice([1, 3, 8, 20]), n_reads=1200,
                                       read_len=rng.choice([60, 100, 150]), snp_every=rng.choice([0, 90, 200]) or 10**9, label_alt=bool(seed % 2))
        cfg = capi.config_cli(k)
        if seed % 5 == 0: cfg.min_seed_length = max(5, k - 6)
        if seed % 7 == 0: cfg.min_exact_match = 0.9
        if seed % 11 == 0: cfg.xdrop = 10
        o = orc.LabeledAlignRun(g, cfg, anno, reads)
        assert o.error == "", o.error
        want = with_labels(o)
        A = aligner.Aligner(gpu_graph(g), cfg, annotation=gpu_annotation(anno))
        for opt in ("lane=1", "seed_lane=1", "ext64=0"):
            A.set_pipeline(opt)
        got, status = A.align_batch(reads)
        assert all(s == 0 for s in status), (seed, [s for s in status if s][:5])
        for q in range(len(reads)):
            assert got[q] == want[q], ("MISMATCH", seed, q, reads[q], got[q], want[q])
        st = A.stats()
        worlds += 1; reads_total += len(reads); lane_total += st["n_lane_reads"]
        A.close()
        seed += 1
        continue
    if seed % 3 == 0:
        k = 31
        g, reads = bench_like_world(seed, 6000, genome_len=rng.choice([60000, 150000]), read_len=rng.choice([100, 150, 150, 250]),
                                    snp_every=rng.choice([0, 490, 120, 60]))
    else:
        k = rng.choice([11, 15, 21, 27, 31])
        g, reads = make_world(seed, k, genome_len=rng.choice([3000, 9000, 30000]), n_reads=1500, read_len=rng.choice([60, 100, 150, 230]),
                              n_variants=rng.choice([0, 10, 40]), mask=seed % 7 == 3)
        reads = [mutate(rng, x, sub=0.03, ins=0.01, dele=0.01) if i % 3 == 0 else x for i, x in enumerate(reads)]
        reads += ["", "ACGT", "N" * 80, "A" * 90, reads[0][:40]]
    cfg = capi.config_cli(k)
    v = rng.randrange(8)
    if v == 1: cfg.left_end_bonus, cfg.right_end_bonus = 2, 3
    elif v == 2: cfg.xdrop = rng.choice([10, 40, 60])
    elif v == 3: cfg.min_seed_length = max(5, k - rng.randrange(1, 12)); cfg.min_exact_match = 0.0
    elif v == 4: capi.set_unit_matrix(cfg, 1); cfg.gap_opening_penalty = -1; cfg.gap_extension_penalty = -1
    elif v == 5: cfg.min_exact_match = rng.choice([0.0, 0.9]); cfg.allow_left_trim = 0
    elif v == 6: cfg.forward_and_reverse_complement = 0; cfg.min_exact_match = 0.0
    want = orc.AlignRun(g, cfg, reads, threads=os.cpu_count() or 8, validate=False).results()
    A = aligner.Aligner(gpu_graph(g), cfg)
    for opt in ("lane=1", "seed_lane=1", "ext64=0"):
        A.set_pipeline(opt)
    got, status = A.align_batch(reads)
    assert all(s == 0 for s in status), (seed, [s for s in status if s][:5])
    for q in range(len(reads)):
        assert got[q] == want[q], ("MISMATCH", seed, q, reads[q], got[q], want[q])
    st = A.stats()
    worlds += 1; reads_total += len(reads); lane_total += st["n_lane_reads"]
    A.close()
    seed += 1
print("ok%s: %d worlds (seeds from %d), %d reads, %d finished by k_lane, no difference" % (" (label-aware)" if labels else "", worlds, first, reads_total, lane_total))

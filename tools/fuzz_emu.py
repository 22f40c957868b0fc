#!/usr/bin/env python3
"""Differential fuzzing of the kernels' wave programs (host model, tests/emu) against the oracle: random graphs in all three
modes, random aligner configurations, random reads; stops at the first difference and prints a reproducer.
    python tools/fuzz_emu.py [--minutes M] [--seed S] [--lanes 64|8]
CPU only (test infrastructure: the oracle is the checker, the host model runs the same sources the HIP build compiles)."""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--lanes", default="")
    ap.add_argument("--start", type=int, default=0, help="first world of the seed's sequence (to re-run a reported one)")
    ap.add_argument("--worlds", type=int, default=0, help="stop after this many worlds (0: by time)")
    ap.add_argument("--verbose", action="store_true", help="print every world before it runs (to find a slow or hanging one)")
    ap.add_argument("--modes", default="0,0,1,2,2", help="graph modes to draw from (0 BASIC, 1 CANONICAL, 2 PRIMARY)")
    ap.add_argument("--lane", action="store_true", help="campaign of the lane-per-read path: BASIC graphs, k <= 32, one alignment per "
                    "query, the split pipeline with MGX_EMU_LANE=1, short reads, more reads per world, scoring variants")
    ap.add_argument("--seedlane", action="store_true", help="campaign of the lane-per-read seeder: BASIC / CANONICAL graphs, k <= 32, the split "
                    "pipeline with MGX_EMU_SEEDLANE=1, reads of up to 160 characters mostly, seed lists compared")
    ap.add_argument("--labels", action="store_true", help="campaign of label-aware alignment (LabeledAligner): BASIC graphs built from "
                    "several diverged strains, a label per strain and per genome segment, compared with the oracle's LabeledAligner "
                    "(alignment lists and their label sets)")
    args = ap.parse_args()
    if args.lanes in ("8", "16"):
        os.environ["MGX_EMU_WAVE"] = args.lanes
    import emu_drv
    import orc
    from metagraph_amd import capi
    from test_emu_vs_oracle import rand_seq, mutate, rc
    from test_oracle_primary_goldens import primary_contigs

    t_end = time.time() + 60 * args.minutes
    it = args.start
    n_reads_total = 0
    n_lane_total = 0
    sl_why = {}
    lab_hist = [0] * 12                  # --labels: reads by their number of alignments (0..5+), alignments by their number of labels
    while time.time() < t_end and not (args.worlds and it >= args.start + args.worlds):
        seed = args.seed * 1000003 + it
        rng = random.Random(seed)
        it += 1
        mode = rng.choice([int(m) for m in args.modes.split(",")])
        k = rng.choice([5, 6, 7, 8, 11, 12, 15, 19, 20, 27, 31, 32, 33, 40, 64])
        if args.lane:
            mode = 0
            k = rng.choice([7, 8, 11, 12, 15, 19, 20, 27, 31, 32])
        if args.labels:
            mode = rng.choice([0, 0, 1, 2])
            k = rng.choice([5, 7, 8, 11, 12, 15, 19, 20, 31, 33])
        if args.seedlane:
            mode = rng.choice([0, 0, 0, 1])
            k = rng.choice([5, 7, 8, 11, 12, 15, 19, 20, 27, 31, 32])
        mask = rng.random() < 0.4
        glen = rng.choice([300, 1000, 3000, 6000])
        genome = rand_seq(rng, glen)
        seqs = [genome]
        if rng.random() < 0.5:                                   # repeats and inverted repeats
            unit = rand_seq(rng, rng.choice([15, 40, 90]))
            genome = genome[:glen // 3] + unit + genome[glen // 3:2 * glen // 3] + (rc(unit) if rng.random() < 0.5 else unit) + genome[2 * glen // 3:]
            seqs = [genome]
        if args.seedlane and rng.random() < 0.6:                 # low-complexity islands: what the DUST filter masks
            g2 = list(genome)
            for _ in range(rng.choice([1, 3, 8])):
                a = rng.randrange(0, max(1, len(g2) - 60))
                u = rng.choice(["A", "C", "G", "T", "AT", "CG", "AC", "GT", "AAT", "CAG", "ACGT", "AACCT"])
                for x in range(a, min(len(g2), a + rng.randint(5, 45))):
                    g2[x] = u[(x - a) % len(u)]
            genome = "".join(g2)
            seqs = [genome]
        for _ in range(rng.choice([0, 5, 30])):                  # variants (bubbles)
            p = rng.randrange(k, len(genome) - k)
            alt = rng.choice([c for c in "ACGT" if c != genome[p]])
            seqs.append(genome[max(0, p - k + 1):p] + alt + genome[p + 1:p + k])
        label_seqs = []                                          # (sequence, label) pairs of the annotation
        if args.labels:
            # strains: copies of the genome with SNPs / indels, one label each; plus labels on segments of the genome (overlapping
            # the strain labels); some sequence in the graph carries no label at all
            n_strains = rng.choice([1, 2, 3, 6])
            for sidx in range(n_strains):
                st = genome if sidx == 0 else mutate(rng, genome, rng.choice([0.005, 0.02, 0.05]))
                if len(st) <= k + 2:
                    st = genome
                seqs.append(st)
                label_seqs.append((st, sidx))
            n_lab = n_strains
            for _ in range(rng.choice([0, 2, 5])):
                a = rng.randrange(0, max(1, len(genome) - 2 * k))
                b = min(len(genome), a + rng.choice([2 * k, 100, 400]))
                label_seqs.append((genome[a:b], n_lab))
                n_lab += 1
            if rng.random() < 0.3:                               # an unlabeled contig sharing a stretch with the genome
                a = rng.randrange(0, max(1, len(genome) - 3 * k))
                seqs.append(rand_seq(rng, 2 * k) + genome[a:a + 3 * k] + rand_seq(rng, 2 * k))
        if mode == 2:
            seqs = primary_contigs(seqs, k, rng.choice(["input", "lex", "colex"]))[0]
        try:
            g = orc.Graph.build(k, seqs, mode, mask)
        except Exception as e:                                   # e.g. a primary set that is not one (palindromic overlap)
            print("skip build:", e)
            continue
        eg = emu_drv.EmuGraph(g, mode=mode)
        anno = ea = None
        if args.labels:
            anno = orc.Annotation(g, max(1, max(l for _, l in label_seqs) + 1))
            for sq, lbl in label_seqs:
                anno.annotate(sq, lbl)
            ea = emu_drv.EmuAnnotation(anno)
        cfg = capi.config_cli(k)
        cfg.min_exact_match = rng.choice([0.0, 0.0, 0.7])
        if rng.random() < 0.5:
            cfg.min_seed_length = rng.randrange(max(3, k // 3), k + 1)
        if rng.random() < 0.3:
            cfg.max_seed_length = rng.choice([k, k + 20, 2 ** 32])
            if cfg.max_seed_length < cfg.min_seed_length:
                cfg.max_seed_length = cfg.min_seed_length
        if args.seedlane and rng.random() < 0.35:
            cfg.max_seed_length = k                              # one seed per matched k-mer (what label-aware alignment sets)
            if cfg.min_seed_length > k:
                cfg.min_seed_length = k
        cfg.max_num_seeds_per_locus = rng.choice([1, 2, 1000])
        cfg.xdrop = rng.choice([10, 27, 27, 50])
        cfg.num_alternative_paths = rng.choice([1, 1, 1, 2, 3])
        if args.lane:
            cfg.num_alternative_paths = 1
            cfg.xdrop = rng.choice([10, 27, 27, 50, 80])
            if rng.random() < 0.3:
                cfg.left_end_bonus, cfg.right_end_bonus = rng.choice([0, 2, 5]), rng.choice([0, 3, 5])
            if rng.random() < 0.2:
                cfg.allow_left_trim = 0
            if rng.random() < 0.2:
                cfg.min_path_score = rng.choice([20, 60, 150])
        cfg.forward_and_reverse_complement = rng.choice([0, 1, 1])
        if rng.random() < 0.3:
            capi.set_dna_matrix(cfg, 2, -rng.choice([1, 3]), -rng.choice([2, 3]))
            cfg.gap_opening_penalty, cfg.gap_extension_penalty = -rng.choice([3, 5]), -rng.choice([1, 2])
        cfg.seed_complexity_filter = rng.choice([0, 1])
        cfg.max_nodes_per_seq_char = rng.choice([5.0, 12.5, 50.0])
        reads = []
        for i in range(rng.choice([5, 20]) if not args.lane else rng.choice([20, 80])):
            L = rng.choice([k - 1, k, k + 3, 40, 100, 150, 150, 400, 1200])      # long reads: wide bands, many seeds, long chains
            if args.lane:
                L = rng.choice([k, k + 3, 40, 75, 100, 150, 150, 150, 250, 256, 257])
            if args.seedlane:
                L = rng.choice([k - 1, k, k + 1, k + 3, 40, 75, 100, 150, 150, 150, 159, 160, 161, 200, 250, 254, 255, 256, 300])
            if k < 11 and L > 150:
                L = 150                                          # (tiny k x long reads: thousands of extensions per read, minutes per world)
            L = max(1, min(L, len(genome) - 1))
            if rng.random() < 0.1:
                r = rand_seq(rng, L)
            else:
                p = rng.randrange(0, len(genome) - L)
                r = mutate(rng, genome[p:p + L], rng.choice([0.0, 0.02, 0.08]))
            if rng.random() < 0.5:
                r = rc(r)
            if rng.random() < 0.1 and len(r) > 4:
                q = rng.randrange(len(r))
                r = r[:q] + rng.choice(["N", "n", "a", "x"]) + r[q + 1:]
            if r:
                reads.append(r)
        for v in ("MGX_EMU_SPLIT", "MGX_EMU_MULTIPASS", "MGX_NO_FAST", "MGX_EMU_LDS", "MGX_EMU_RESUME_CAP", "MGX_EMU_LABEL_SCALE"):
            os.environ.pop(v, None)
        if rng.random() < 0.5:                                   # modelled LDS size: which per-read arrays live in LDS / in the arena
            os.environ["MGX_EMU_LDS"] = str(rng.choice([0, 256, 700, 1500, 4000, 20000]))
        if rng.random() < 0.5:
            os.environ["MGX_EMU_SPLIT"] = "1"
            if rng.random() < 0.5:
                os.environ["MGX_EMU_MULTIPASS"] = "1"
                if rng.random() < 0.3:
                    os.environ["MGX_EMU_RESUME_CAP"] = str(rng.choice([1, 2, 5]))      # resume-record pool smaller than the batch
        if rng.random() < 0.15:
            os.environ["MGX_NO_FAST"] = "1"
        os.environ.pop("MGX_EMU_LANE", None)
        if args.lane:
            os.environ.pop("MGX_NO_FAST", None)
            os.environ.pop("MGX_EMU_MULTIPASS", None)
            os.environ["MGX_EMU_SPLIT"] = "1"
            os.environ["MGX_EMU_LANE"] = "1"
        os.environ.pop("MGX_EMU_SEEDLANE", None)
        if args.seedlane:
            os.environ["MGX_EMU_SPLIT"] = "1"
            os.environ["MGX_EMU_SEEDLANE"] = "1"
        if args.labels:
            os.environ.pop("MGX_EMU_MULTIPASS", None)
            cfg.num_alternative_paths = rng.choice([1, 1, 1, 2, 3, 4])
            if rng.random() < 0.25:                                # the label arenas as the capacity retry of mgx_align_batch sizes them
                os.environ["MGX_EMU_LABEL_SCALE"] = str(rng.choice([2, 4]))
            if rng.random() < 0.3:
                cfg.left_end_bonus, cfg.right_end_bonus = rng.choice([0, 2, 5]), rng.choice([0, 3, 5])
        desc = dict(seed=seed, mode=mode, k=k, mask=mask, glen=len(genome), n_seqs=len(seqs), msl=cfg.min_seed_length,
                    maxsl=cfg.max_seed_length, per_locus=cfg.max_num_seeds_per_locus, xdrop=cfg.xdrop, n_alt=cfg.num_alternative_paths,
                    fwd_rc=cfg.forward_and_reverse_complement, mem=cfg.min_exact_match,
                    env={v: os.environ.get(v) for v in ("MGX_EMU_SPLIT", "MGX_EMU_MULTIPASS", "MGX_NO_FAST", "MGX_EMU_LDS", "MGX_EMU_RESUME_CAP",
                                                         "MGX_EMU_LABEL_SCALE")})
        if args.verbose:
            print(desc, [len(r) for r in reads], flush=True)
        try:
            t_w = time.time()
            # (end bonuses: the reference's own Alignment::is_valid — a debug assertion there — rejects some alignments it
            # produces; the comparison is against what it produces)
            if args.labels:
                o = orc.LabeledAlignRun(g, cfg, anno, reads, validate=not (cfg.left_end_bonus or cfg.right_end_bonus))
            else:
                o = orc.AlignRun(g, cfg, reads, validate=not (cfg.left_end_bonus or cfg.right_end_bonus))
            if args.verbose:
                print('  oracle %.1fs' % (time.time() - t_w), flush=True)
            if o.error:
                print("oracle error (skipped):", o.error[:100], desc)
                continue
            e = emu_drv.EmuRun(eg, cfg, reads, annotation=ea)
            assert e.error == "", e.error
            got, status = e.results()
            want = o.results()
            if args.labels:
                for q, per_aln in enumerate(o.labels()):
                    for a, ls in zip(want[q], per_aln):
                        a["labels"] = [int(x) for x in ls]
                n_cap = sum(1 for st in status if st != 0)
                if n_cap:
                    print("  capacity:", n_cap, "of", len(reads), desc)
                for q in range(len(reads)):
                    lab_hist[min(len(want[q]), 5)] += 1
                    for a in want[q]:
                        lab_hist[6 + min(len(a["labels"]), 5)] += 1
            for q in range(len(reads)):
                if status[q] != 0:
                    continue                                     # capacity status: allowed, never a wrong answer
                assert got[q] == want[q], ("read %d %s" % (q, reads[q]), got[q], want[q])
            if all(st == 0 for st in status):                    # the seeders' products too (seed lists, num_matches per strand)
                info = e.seed_info()
                for strand in (0, 1):
                    for q, (ss, nm) in enumerate(o.seeds(strand)):
                        assert info[q]["num_matches"][strand] == nm, ("num_matches", q, strand, reads[q])
                        assert info[q]["seeds"][strand] == emu_drv.oracle_seeds_as_tuples(ss), ("seeds", q, strand, reads[q], info[q]["seeds"][strand], emu_drv.oracle_seeds_as_tuples(ss))
            n_reads_total += len(reads)
            if args.lane:
                n_lane_total += e.lane_stats()[1]
            if args.seedlane:
                ran, done, why = e.seedlane_stats()
                n_lane_total += done
                for c, v in why.items():
                    sl_why[c] = sl_why.get(c, 0) + v
        except AssertionError as ex:
            print("MISMATCH", desc)
            print(str(ex)[:3000])
            sys.exit(1)
    print("ok: %d worlds, %d reads, no difference" % (it, n_reads_total) + (" (%d reads finished by the lane path)" % n_lane_total if args.lane else "")
          + (" (%d reads seeded by the lane-per-read seeder; left by reason: %s)" % (n_lane_total, dict(sorted(sl_why.items()))) if args.seedlane else "")
          + (" (reads with 0..5+ alignments: %s; alignments with 0..5+ labels: %s)" % (lab_hist[:6], lab_hist[6:]) if args.labels else ""))


if __name__ == "__main__":
    main()

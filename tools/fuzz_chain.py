#!/usr/bin/env python3
"""Differential fuzzing of post-alignment chaining (config.post_chain_alignments): random genomes, reads stitched from two or three
places (gaps, overlaps, errors, both strands), the kernels' host model with the keep-every-alignment aggregator + libmgx's
host-side chaining (csrc/chain_host.hpp, through mgx_chain_alignments) against the oracle's chain_alignments.  CPU only.
    python tools/fuzz_chain.py [--minutes M] [--seed S]"""
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
    args = ap.parse_args()
    import orc
    from metagraph_amd import capi
    from test_emu_vs_oracle import rand_seq, mutate, rc
    from test_oracle_chain import chain_config, product_chain
    t_end = time.time() + 60 * args.minutes
    it = n_reads = n_chained = n_capacity = 0
    while time.time() < t_end:
        seed = args.seed * 1000003 + it
        it += 1
        rng = random.Random(seed)
        k = rng.choice([9, 10, 12, 15, 19, 21, 31])
        glen = rng.choice([800, 2000, 5000])
        genome = rand_seq(rng, glen)
        seqs = [genome]
        if rng.random() < 0.4:                      # a second, diverged copy: bubbles
            seqs.append(mutate(rng, genome, 0.02))
        g = orc.Graph.build(k, seqs, 0, rng.random() < 0.3)
        scores = rng.choice([(2, -1, -2), (2, -3, -3), (1, -1, -1)])
        cfg = chain_config(k, scores, (-1, -1) if scores[0] == 1 else None)
        cfg.min_seed_length = rng.choice([k, k, max(8, k - 4)])
        cfg.num_alternative_paths = rng.choice([1, 1, 2])
        if rng.random() < 0.3:
            cfg.forward_and_reverse_complement = 0
        queries = []
        for _ in range(20):
            parts = []
            for _p in range(rng.choice([2, 2, 3])):
                a = rng.randrange(0, glen - 80)
                parts.append(genome[a:a + rng.randrange(k + 3, 70)])
                parts.append(rand_seq(rng, rng.choice([0, 0, 0, 1, 2, 6, 15])))
            q = "".join(parts)
            if rng.random() < 0.2:                  # overlapping pieces: the second starts inside the first
                a = rng.randrange(0, glen - 120)
                ov = rng.randrange(1, k + 5)
                q = genome[a:a + 50] + genome[a + 50 - ov + rng.choice([200, -150]) % (glen - 120):][:50]
            if rng.random() < 0.5:
                q = mutate(rng, q, 0.02)
            if rng.random() < 0.4:
                q = rc(q)
            queries.append(q[:200])
        try:
            want = orc.AlignRun(g, cfg, queries).results()
            plain, got = product_chain(g, k, cfg, queries)
        except Exception as e:                                        # noqa: BLE001
            print("world", seed, "k", k, "raised", repr(e)[:300])
            raise
        for q in range(len(queries)):
            n_reads += 1
            if got[q] is None:
                n_capacity += 1
                continue
            if got[q] != want[q]:
                print("MISMATCH world", seed, "k", k, "query", q, queries[q], "\nplain", plain[q], "\ngot", got[q], "\nwant", want[q])
                sys.exit(1)
            n_chained += any(0 in a["nodes"] for a in got[q])
    print("fuzz_chain: %d worlds, %d reads, %d with a chain among their alignments, %d capacity statuses, no difference"
          % (it, n_reads, n_chained, n_capacity))


if __name__ == "__main__":
    main()

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
    ap.add_argument("--world-seconds", type=int, default=60)
    args = ap.parse_args()
    import orc
    import emu_drv  # noqa: F401  (builds the host model if it is stale: here, not inside a world's time limit)
    from metagraph_amd import capi
    capi.lib()                                  # (loaded here once: not by every world's child, while a rebuild may be replacing it)
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    t_end = time.time() + 60 * args.minutes
    it = n_reads = n_chained = n_capacity = n_slow = 0
    while time.time() < t_end:
        seed = args.seed * 1000003 + it
        it += 1
        # One world per child process: the plain oracle's extension is exponential-ish on small k with cheap gaps (as the
        # reference's is); a world it does not finish in --world-seconds is skipped and counted, not waited for.
        rd, wr = ctx.Pipe(False)
        pr = ctx.Process(target=world, args=(seed, wr, orc, capi))
        pr.start()
        wr.close()
        res, oracle_done = None, False
        t_world = time.time() + args.world_seconds
        while res is None and rd.poll(max(0.0, t_world - time.time())):
            msg = rd.recv()
            if msg[0] == "oracle_done":
                oracle_done = True
            else:
                res = msg
        if res is None:
            pr.terminate()
        pr.join()
        if res is None:
            if oracle_done:                     # the oracle finished, the product side did not: that is a finding
                print("SLOW PRODUCT world", seed, "(oracle done, host model + chaining not within %d s)" % args.world_seconds)
                sys.exit(1)
            n_slow += 1
            continue
        if res[0] == "mismatch":
            print(res[1])
            sys.exit(1)
        n_reads += res[1]; n_chained += res[2]; n_capacity += res[3]
    print("fuzz_chain: %d worlds (%d skipped: oracle slower than %d s), %d reads, %d with a chain among their alignments, "
          "%d capacity statuses, no difference" % (it, n_slow, args.world_seconds, n_reads, n_chained, n_capacity))


def world(seed, wr, orc, capi):
    from test_emu_vs_oracle import rand_seq, mutate, rc
    from test_oracle_chain import chain_config, product_chain
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
    want = orc.AlignRun(g, cfg, queries).results()
    wr.send(("oracle_done",))
    plain, got = product_chain(g, k, cfg, queries)
    n_reads = n_chained = n_capacity = 0
    for q in range(len(queries)):
        n_reads += 1
        if got[q] is None:
            n_capacity += 1
            continue
        if got[q] != want[q]:
            wr.send(("mismatch", "MISMATCH world %d k %d query %d %s\nplain %s\ngot %s\nwant %s"
                     % (seed, k, q, queries[q], plain[q], got[q], want[q])))
            return
        n_chained += any(0 in a["nodes"] for a in got[q])
    wr.send(("ok", n_reads, n_chained, n_capacity))


if __name__ == "__main__":
    main()

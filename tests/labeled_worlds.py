"""Random worlds for label-aware alignment (TEST INFRASTRUCTURE): a BASIC graph over several diverged strains of one genome,
a label per strain plus labels on genome segments (so that nodes carry 0, 1 or many labels and label sets change along
paths), reads drawn from the strains."""
import random

import orc
from test_emu_vs_oracle import rand_seq, mutate, rc


def labeled_world(seed, k, n_strains=3, genome_len=1500, n_reads=20, read_len=100, n_segments=3, divergence=0.02, mask=True,
                  unlabeled_contig=True):
    rng = random.Random(seed)
    genome = rand_seq(rng, genome_len)
    strains = [genome]
    for _ in range(n_strains - 1):
        st = mutate(rng, genome, divergence)
        strains.append(st if len(st) > k + 2 else genome)
    seqs = list(strains)
    label_seqs = [(st, j) for j, st in enumerate(strains)]
    n_lab = n_strains
    for _ in range(n_segments):
        a = rng.randrange(0, max(1, genome_len - 2 * k))
        label_seqs.append((genome[a:a + rng.choice([2 * k, 100, 400])], n_lab))
        n_lab += 1
    if unlabeled_contig:
        a = rng.randrange(0, max(1, genome_len - 3 * k))
        seqs.append(rand_seq(rng, 2 * k) + genome[a:a + 3 * k] + rand_seq(rng, 2 * k))
    g = orc.Graph.build(k, seqs, 0, mask)
    anno = orc.Annotation(g, n_lab)
    for sq, lbl in label_seqs:
        if len(sq) >= k:
            anno.annotate(sq, lbl)
    reads = []
    for i in range(n_reads):
        if i % 11 == 10:
            reads.append(rand_seq(rng, read_len))
            continue
        st = rng.choice(strains)
        L = min(read_len, len(st) - 1)
        p = rng.randrange(0, len(st) - L)
        r = mutate(rng, st[p:p + L], rng.choice([0.0, 0.02, 0.06]))
        if rng.random() < 0.5:
            r = rc(r)
        if r:
            reads.append(r)
    return g, anno, reads


def with_labels(run):
    """orc.LabeledAlignRun -> results() with every alignment's "labels" filled in (the form capi.results_to_py gives)"""
    want = run.results()
    for q, per_aln in enumerate(run.labels()):
        for a, ls in zip(want[q], per_aln):
            a["labels"] = [int(x) for x in ls]
    return want

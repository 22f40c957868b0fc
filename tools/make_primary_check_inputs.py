#!/usr/bin/env python3
"""Inputs and expected outputs of tools/check_primary_on_gpu.sh -> gpurun_in/ (git-ignored; travels to the GPU box with the
snapshot): BOSS dumps of primary graphs, reads, and the oracle's TSV lines for them.  CPU only; the oracle is the checker."""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import orc  # noqa: E402
from metagraph_amd import capi  # noqa: E402
from test_oracle_kats import read_fasta, read_fastq, HERE  # noqa: E402
from test_oracle_primary_goldens import primary_contigs, PRIMARY  # noqa: E402
from test_emu_primary import primary_world  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_in")


def dump(g, path):
    W, last, F, _ = g.export()
    with open(path, "wb") as f:
        f.write(struct.pack("<7Q", g.k, g.n_edges, *[int(x) for x in F]))
        f.write(W.tobytes())
        f.write(last.tobytes())


def main():
    os.makedirs(OUT, exist_ok=True)
    contigs, _ = primary_contigs(read_fasta(os.path.join(HERE, "golden", "genome.MT.fa")), 11)
    g = orc.Graph.build(11, contigs, PRIMARY, False)
    dump(g, os.path.join(OUT, "mt.primary.boss"))
    reads = read_fastq(os.path.join(HERE, "golden", "genome_MT1.fq"))
    for tag, msl in (("a", None), ("b", 10)):
        cfg = capi.config_cli(11)
        cfg.min_exact_match = 0.0
        if msl:
            cfg.min_seed_length = msl
        lines = orc.AlignRun(g, cfg, [r[1] for r in reads]).tsv_lines()
        with open(os.path.join(OUT, "mt.expect.%s.tsv" % tag), "w") as f:
            f.write("".join(reads[i][0] + l[l.index("\t"):] + "\n" for i, l in enumerate(lines)))
    worlds = (("w31", 31, 703, "colex", None, None, 20000, 2000, 100), ("w12", 12, 705, "lex", None, 0.0, 8000, 1500, 80),
              ("w15", 15, 750, "input", 9, 0.0, 8000, 1500, 80), ("w20", 20, 761, "colex", 12, None, 8000, 1500, 80))
    for name, k, seed, order, msl, mem, glen, n_reads, n_var in worlds:
        g2, rd = primary_world(seed, k, genome_len=glen, n_reads=n_reads, n_variants=n_var, order=order)
        dump(g2, os.path.join(OUT, "%s.primary.boss" % name))
        with open(os.path.join(OUT, "%s.fa" % name), "w") as f:
            f.write("".join(">r%d\n%s\n" % (i, r) for i, r in enumerate(rd)))
        cfg = capi.config_cli(k)
        if msl:
            cfg.min_seed_length = msl
        if mem is not None:
            cfg.min_exact_match = mem
        lines = orc.AlignRun(g2, cfg, rd).tsv_lines()
        with open(os.path.join(OUT, "%s.expect.tsv" % name), "w") as f:
            f.write("".join("r%d" % i + l[l.index("\t"):] + "\n" for i, l in enumerate(lines)))
        print(name, len(lines), "reads")


if __name__ == "__main__":
    main()

"""The reference's own files as inputs of the device path (mgx_graph_load_dbg, mgx_annotation_create_from_file; the driver's
`.dbg` / `.column.annodbg` arguments): the graph and annotation the reference wrote (examples/data/graphs) aligned on the GPU
against the oracle over the same sequences, then larger tables in every readable state written by tests/sdsl_writer.py.  Needs
a real MI355X."""
import os
import subprocess

import numpy as np
import pytest

import orc
import sdsl_writer as sw
from metagraph_amd import aligner, capi
from labeled_worlds import with_labels
from test_boss_files import GOLD, read_fasta
from test_emu_primary import primary_world
from test_gpu_parity import compare_gpu

pytestmark = pytest.mark.gpu

EXE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "metagraph_amd", "_build", "mgx_align")


def dna_world():
    seqs = read_fasta(os.path.join(GOLD, "test_DNA_sequences.fa"))
    g = orc.Graph.build(20, seqs, 0, False)
    anno = orc.Annotation(g, 1)
    for s in seqs:
        anno.annotate(s, 0)
    reads = read_fasta(os.path.join(GOLD, "test_DNA_query.fa"))
    rng = np.random.default_rng(4)
    for s in seqs:                                   # + substrings of the annotated sequences with a substitution each
        for _ in range(6):
            a = int(rng.integers(0, len(s) - 30))
            r = list(s[a:a + int(rng.integers(24, 50))])
            p = int(rng.integers(3, len(r) - 3))
            r[p] = "ACGT"[("ACGT".index(r[p]) + 1 + int(rng.integers(0, 3))) % 4]
            reads.append("".join(r))
    return g, anno, reads


def test_reference_written_graph_and_annotation_on_gpu():
    """Graph.load(test_DNA_graph.dbg) + Annotation.load(test_DNA_graph.column.annodbg): label-aware and plain alignment equal
    the oracle's over a graph and an annotation built from test_DNA_sequences.fa"""
    g, anno, reads = dna_world()
    G = aligner.Graph.load(os.path.join(GOLD, "test_DNA_graph.dbg"))
    assert (G.k, G.n_edges, G.mode) == (20, 25, 0)
    AN = aligner.Annotation.load(os.path.join(GOLD, "test_DNA_graph.column.annodbg"))
    assert AN.labels == ["test_DNA_sequences.fa"] and AN.n_rows == 25
    cfg = capi.config_cli(20)
    cfg.min_exact_match = 0.0
    o = orc.LabeledAlignRun(g, cfg, anno, reads)
    assert o.error == "", o.error
    A = aligner.Aligner(G, cfg, annotation=AN)
    got, status = A.align_batch(reads)
    assert all(s == 0 for s in status)
    want = with_labels(o)
    assert sum(len(w) for w in want) >= len(reads) // 2
    for q in range(len(reads)):
        assert got[q] == want[q], (q, reads[q], got[q], want[q])
    compare_gpu(g, G, cfg, reads)                    # and without the annotation: DBGAligner<>


def test_driver_reads_the_reference_written_files(tmp_path):
    """mgx_align GRAPH.dbg READS -a X.column.annodbg: the line `metagraph align -a` prints (cli/align.cpp:274-281)"""
    g, anno, reads = dna_world()
    fa = tmp_path / "q.fa"
    fa.write_text("".join(">r%d\n%s\n" % (i, r) for i, r in enumerate(reads)))
    r = subprocess.run([EXE, os.path.join(GOLD, "test_DNA_graph.dbg"), str(fa), "-a", os.path.join(GOLD, "test_DNA_graph.column.annodbg"),
                        "--align-min-exact-match", "0.0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    cfg = capi.config_cli(20)
    cfg.min_exact_match = 0.0
    o = orc.LabeledAlignRun(g, cfg, anno, reads)
    lines = []
    for i, (res, labs) in enumerate(zip(o.results(), o.labels())):
        line = "r%d\t%s" % (i, reads[i])
        for a, ls in zip(res, labs):
            line += "\t%s\t%s\t%d\t%d\t%s\t%d\t%s" % ("-" if a["orientation"] else "+", a["sequence"], a["score"], a["num_matches"], a["cigar"],
                                                    a["offset"], ";".join("test_DNA_sequences.fa" for _ in ls))
        if not res:
            line += "\t*\t*\t%d\t*\t*\t*" % cfg.min_path_score
        lines.append(line)
    assert r.stdout.splitlines() == lines
    # a protein graph is refused with the reason, not aligned
    r = subprocess.run([EXE, os.path.join(GOLD, "test_Protein_graph.dbg"), str(fa)], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "alphabet" in r.stderr
    # an annotation over other rows than the graph's nodes
    other = tmp_path / "o.column.annodbg"
    other.write_bytes(sw.column_file(24, ["x"], [[1, 2]]))
    r = subprocess.run([EXE, os.path.join(GOLD, "test_DNA_graph.dbg"), str(fa), "-a", str(other)], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "rows" in r.stderr


@pytest.mark.parametrize("state,last_code,mode", [(sw.STATE_SMALL, sw.CODE_RRR, 0), (sw.STATE_SMALL, sw.CODE_SD, 1), (sw.STATE_STAT, None, 2),
                                                  (sw.STATE_FAST, None, 0)])
def test_written_graph_files_on_gpu(tmp_path, state, last_code, mode):
    """a table of 10 000 - 36 000 edges per readable state and graph mode, written in the reference's layout, loaded by mgx_graph_load_dbg,
    two annotation files side by side: 300 reads equal the oracle's alignments and label lists"""
    rng = np.random.default_rng(20 + state + mode)
    k = 15
    genome = ["".join(rng.choice(list("ACGT"), size=6000)) for _ in range(3)]
    if mode == 2:                                    # PRIMARY: the primary contigs of a genome with variants (test_emu_primary.py)
        g, primary_reads = primary_world(31, k, genome_len=6000, n_reads=300)
    else:
        g = orc.Graph.build(k, genome, mode, False)
    W, last, F, _ = g.export()
    path = tmp_path / "w.dbg"
    path.write_bytes(sw.dbg_file(k, W, last, [int(x) for x in F], mode=mode, state=state, last_code=last_code or sw.CODE_RRR))
    G = aligner.Graph.load(path)
    assert (G.k, G.n_edges, G.mode) == (k, g.n_edges, mode)
    reads = []
    for _ in range(300):
        s = genome[int(rng.integers(0, 3))]
        a = int(rng.integers(0, len(s) - 120))
        r = list(s[a:a + int(rng.integers(40, 120))])
        for p in rng.integers(0, len(r), size=int(rng.integers(0, 4))):
            r[int(p)] = "ACGT"[int(rng.integers(0, 4))]
        reads.append("".join(r))
    cfg = capi.config_cli(k)
    cfg.min_exact_match = 0.0
    if mode == 0:                                    # label-aware: BASIC graphs through both files
        anno = orc.Annotation(g, 3)
        for j, s in enumerate(genome):
            anno.annotate(s, j)
        cols = []
        for j in range(3):
            words = anno.column_words(j)
            cols.append([r for r in range(g.n_edges) if (int(words[r >> 6]) >> (r & 63)) & 1])
        a, b = tmp_path / "a.column.annodbg", tmp_path / "b.column.annodbg"
        a.write_bytes(sw.column_file(g.n_edges, ["g0", "g1"], cols[:2], codes=[sw.CODE_SD, sw.CODE_STAT], v2=True))
        b.write_bytes(sw.column_file(g.n_edges, ["g2"], cols[2:], v2=False))
        AN = aligner.Annotation.load([a, b])
        assert AN.labels == ["g0", "g1", "g2"]
        o = orc.LabeledAlignRun(g, cfg, anno, reads)
        assert o.error == "", o.error
        got, status = aligner.Aligner(G, cfg, annotation=AN).align_batch(reads)
        want = with_labels(o)
        assert all(s == 0 for s in status)
        assert sum(len(w) for w in want) >= 250
        for q in range(len(reads)):
            assert got[q] == want[q], (q, reads[q])
    else:
        compare_gpu(g, G, cfg, primary_reads if mode == 2 else reads)

"""PRIMARY-mode graphs on the GPU through the C-ABI (mgx_boss_view.mode = MGX_MODE_PRIMARY): the CanonicalDBG wrapper of
metagraph_amd/csrc/canon_graph.hpp — wrapper mapping, reverse-complement sub-k seeds, wrapper traversal in the extender — on
the reference's PRIMARY KATs, the primary genome.MT CLI goldens byte for byte (integration_tests/test_align.py:271-329) and
seeded random worlds against the oracle.

Hardware runs in round 2: tools/check_primary_on_gpu.sh — the C++ driver on six primary graphs, 6514 reads, TSV byte-identical
to the oracle's in both wrapper modes (profiles/r02_primary_hw_check.txt) — and this file before the table mode existed
(profiles/r02_primary_gpu_pytest.txt).  The file sorts last on purpose: it is the newest path."""
import ctypes as C
import os
import struct
import subprocess

import pytest

import orc
from metagraph_amd import aligner, capi
from test_oracle_kats import read_fasta, read_fastq, HERE
from test_oracle_canonical import CANONICAL_LINES, SUBK_LINE_5, _cfg
from test_oracle_primary_goldens import primary_contigs, PRIMARY
from test_emu_primary import primary_world

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(HERE)


@pytest.fixture(autouse=True, params=["tables", "no-tables"])
def _wrapper_mode(request, monkeypatch):
    """Both ways the device walks the wrapper: reverse-complement tables built at load time (default) and, with
    MGX_PRIMARY_TABLES=0, spellings and look-ups re-derived per expansion (read by mgx_graph_create)."""
    if request.param == "no-tables":
        monkeypatch.setenv("MGX_PRIMARY_TABLES", "0")
    else:
        monkeypatch.delenv("MGX_PRIMARY_TABLES", raising=False)


def gpu_graph(g):
    W, last, F, valid = g.export()
    return aligner.Graph(g.k, W, last, F, valid, mode=PRIMARY)


def test_primary_kats_on_gpu(kernels):
    g = orc.Graph.build(18, ["TTGGCCTCGAAAGTTTTT"], PRIMARY, False)
    cfg = _cfg(max_num_seeds_per_locus=capi.UINT64_MAX, min_cell_score=-2147483648 + 100, min_path_score=-2147483648 + 100,
               min_seed_length=13)
    got, status = aligner.Aligner(gpu_graph(g), cfg).align_batch(["GGGGGCTTTCGAGGCCAA"])
    assert status == [0] and got == orc.AlignRun(g, cfg, ["GGGGGCTTTCGAGGCCAA"]).results()
    g = orc.Graph.build(31, ["CTGCTGCGCCATCGCAACCCACGGTTGCTTTTTGAGTCGCTGCTCACGTTAGCCATCACACTGACGTTAAGCTGGCTTTCGATGCTGTATC"],
                        PRIMARY, False)
    query = "CTTACTGCTGCGCTCTTCGCAAACCCCACGGTTTCTTGTTTTGAGCTCGCCTGCTCACGATACCCATACACACTGACGTTCAAGCTGGCTTTCGATGTTGTATC"
    for msl in (0, 131):
        cfg = _cfg(max_num_seeds_per_locus=capi.UINT64_MAX, min_cell_score=-2147483648 + 100,
                   min_path_score=-2147483648 + 100, min_seed_length=13, max_seed_length=msl)
        got, status = aligner.Aligner(gpu_graph(g), cfg).align_batch([query])
        assert status == [0] and len(got[0]) == 1 and got == orc.AlignRun(g, cfg, [query]).results()


@pytest.mark.parametrize("min_seed_length", [None, 10])
def test_primary_cli_goldens_on_gpu(min_seed_length, kernels):
    contigs, _ = primary_contigs(read_fasta(os.path.join(HERE, "golden", "genome.MT.fa")), 11)
    g = orc.Graph.build(11, contigs, PRIMARY, False)
    reads = read_fastq(os.path.join(HERE, "golden", "genome_MT1.fq"))
    cfg = capi.config_cli(11)
    cfg.min_exact_match = 0.0
    if min_seed_length is not None:
        cfg.min_seed_length = min_seed_length
    gg = gpu_graph(g)
    assert capi.lib().mgx_graph_max_index(gg.h) == 2 * g.n_edges
    A = aligner.Aligner(gg, cfg)
    blob, offs = aligner.pack_queries([r[1] for r in reads])
    res = capi.Results()
    assert capi.lib().mgx_align_batch(A.h, blob, offs.ctypes.data, len(reads), 0, C.byref(res)) == 0
    lines = [A.format_tsv(res, i, reads[i][0], reads[i][1]).rstrip("\n") for i in range(len(reads))]
    assert len(lines) == 7
    for i, want in CANONICAL_LINES.items():
        assert lines[i] == want
    assert lines[6].split("\t")[4] == "310"
    if min_seed_length == 10:
        assert lines[5] == SUBK_LINE_5
    else:
        last = lines[5].split("\t")
        assert last[0] == "MT-11/1" and last[4] == "22"


@pytest.mark.parametrize("k,mask,seed,order", [(11, False, 1, "input"), (31, False, 3, "colex"), (15, True, 4, "input"),
                                               (12, False, 5, "lex"), (40, False, 7, "lex")])
def test_primary_random_worlds_on_gpu(k, mask, seed, order, kernels):
    g, reads = primary_world(700 + seed, k, mask=mask, order=order, n_reads=200)
    cfg = capi.config_cli(k)
    want = orc.AlignRun(g, cfg, reads).results()
    got, status = aligner.Aligner(gpu_graph(g), cfg).align_batch(reads)
    assert all(s == 0 for s in status)
    assert got == want


def test_primary_alternative_paths_and_sub_k_on_gpu(kernels):
    g, reads = primary_world(750, 15, genome_len=4000, n_reads=200, n_variants=40)
    gg = gpu_graph(g)
    for n_alt, msl in ((1, 15), (2, 11), (1, 9)):
        cfg = capi.config_cli(15)
        cfg.min_exact_match = 0.0
        cfg.num_alternative_paths = n_alt
        cfg.min_seed_length = msl
        got, status = aligner.Aligner(gg, cfg).align_batch(reads)
        assert all(s == 0 for s in status)
        assert got == orc.AlignRun(g, cfg, reads).results()


def test_primary_world_at_full_occupancy_on_gpu():
    """30 000 reads on a PRIMARY graph: more reads than resident groups, so the 8-lane PRIMARY product kernel
    (k_align_grp8_prim) runs 8 reads per wavefront at full occupancy, as the benchmark's --graph-mode primary does; every
    read against the oracle."""
    g, reads = primary_world(760, 31, genome_len=120000, n_reads=30000, read_len=150, n_variants=600)
    cfg = capi.config_cli(31)
    A = aligner.Aligner(gpu_graph(g), cfg)
    got, status = A.align_batch(reads)
    assert all(s == 0 for s in status)
    assert A.stats()["extend_kernels"] & capi.KERNEL_GRP8_PRIM
    want = orc.AlignRun(g, cfg, reads, threads=os.cpu_count() or 8, validate=False).results()
    assert got == want


def test_mgx_align_driver_on_a_primary_graph(tmp_path):
    contigs, _ = primary_contigs(read_fasta(os.path.join(HERE, "golden", "genome.MT.fa")), 11)
    g = orc.Graph.build(11, contigs, PRIMARY, False)
    W, last, F, _ = g.export()
    dump = tmp_path / "mt.primary.boss"
    with open(dump, "wb") as f:
        f.write(struct.pack("<7Q", g.k, g.n_edges, *[int(x) for x in F]))
        f.write(W.tobytes())
        f.write(last.tobytes())
    exe = os.path.join(ROOT, "metagraph_amd", "_build", "mgx_align")
    reads = os.path.join(HERE, "golden", "genome_MT1.fq")
    for extra, line5 in (([], None), (["--align-min-seed-length", "10"], SUBK_LINE_5)):
        r = subprocess.run([exe, str(dump), reads, "--primary", "--align-min-exact-match", "0.0"] + extra,
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.rstrip("\n").split("\n")
        assert len(lines) == 7
        for i, want in CANONICAL_LINES.items():
            assert lines[i] == want
        if line5:
            assert lines[5] == line5

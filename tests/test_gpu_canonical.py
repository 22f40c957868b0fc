"""CANONICAL-mode DBGSuccinct graphs on the GPU through the C-ABI (mgx_boss_view.mode = MGX_MODE_CANONICAL): the reference's
canonical KATs, the genome.MT canonical CLI goldens byte for byte through mgx_format_tsv, and seeded random worlds against the
oracle.  (PRIMARY graphs: tests/test_gpu_zz_primary.py.)"""
import ctypes as C
import os

import pytest

import orc
from metagraph_amd import aligner, capi
from test_oracle_kats import read_fasta, read_fastq, HERE
from test_oracle_canonical import CANONICAL, CANONICAL_LINES, SUBK_LINE_5, _cfg
from test_emu_canonical import canonical_world

pytestmark = pytest.mark.gpu


def gpu_graph(g, mode=CANONICAL):
    W, last, F, valid = g.export()
    return aligner.Graph(g.k, W, last, F, valid, mode=mode)


def test_canonical_kats_on_gpu(kernels):
    g = orc.Graph.build(7, ["AAAAGCTTTCGAGGCCAA"], CANONICAL, True)
    cfg = _cfg()
    got, status = aligner.Aligner(gpu_graph(g), cfg).align_batch(["AAAAGTTTTCGAGGCCAA"])
    assert status == [0] and got == orc.AlignRun(g, cfg, ["AAAAGTTTTCGAGGCCAA"]).results()
    assert got[0][0]["cigar"] in ("5=1X12=", "12=1X5=") and len(got[0][0]["nodes"]) == 12
    g = orc.Graph.build(18, ["TTGGCCTCGAAAGTTTTT"], CANONICAL, False)
    cfg = _cfg(max_num_seeds_per_locus=capi.UINT64_MAX, min_cell_score=-2147483648 + 100, min_path_score=-2147483648 + 100,
               min_seed_length=13)
    got, status = aligner.Aligner(gpu_graph(g), cfg).align_batch(["GGGGGCTTTCGAGGCCAA"])
    assert status == [0] and got == orc.AlignRun(g, cfg, ["GGGGGCTTTCGAGGCCAA"]).results()
    assert got[0][0]["offset"] == 5 and got[0][0]["num_matches"] == 13


@pytest.mark.parametrize("min_seed_length", [None, 10])
def test_canonical_cli_goldens_on_gpu(min_seed_length, kernels):
    # integration_tests/test_align.py:209-268
    g = orc.Graph.build(11, read_fasta(os.path.join(HERE, "golden", "genome.MT.fa")), CANONICAL, True)
    reads = read_fastq(os.path.join(HERE, "golden", "genome_MT1.fq"))
    cfg = capi.config_cli(11)
    cfg.min_exact_match = 0.0
    if min_seed_length is not None:
        cfg.min_seed_length = min_seed_length
    A = aligner.Aligner(gpu_graph(g), cfg)
    blob, offs = aligner.pack_queries([r[1] for r in reads])
    res = capi.Results()
    assert capi.lib().mgx_align_batch(A.h, blob, offs.ctypes.data, len(reads), 0, C.byref(res)) == 0
    lines = [A.format_tsv(res, i, reads[i][0], reads[i][1]).rstrip("\n") for i in range(len(reads))]
    assert len(lines) == 7
    for i, want in CANONICAL_LINES.items():
        assert lines[i] == want
    if min_seed_length == 10:
        assert lines[5] == SUBK_LINE_5
    else:
        last = lines[5].split("\t")
        assert last[0] == "MT-11/1" and last[4] == "22"


@pytest.mark.parametrize("k,mask,seed", [(11, False, 1), (31, False, 3), (15, True, 4)])
def test_canonical_random_worlds_on_gpu(k, mask, seed, kernels):
    g, reads = canonical_world(900 + seed, k, mask=mask, n_reads=200)
    cfg = capi.config_cli(k)
    want = orc.AlignRun(g, cfg, reads).results()
    got, status = aligner.Aligner(gpu_graph(g), cfg).align_batch(reads)
    assert all(s == 0 for s in status)
    assert got == want
    assert all(a["orientation"] == 0 for q in got for a in q)


def test_primary_graphs_beyond_k_64_are_refused():
    """PRIMARY graphs run through the CanonicalDBG wrapper (tests/test_gpu_zz_primary.py); the kernels that build its tables hold
    node spellings in registers, so k > 64 is refused loudly."""
    seq = "AAAAGCTTTCGAGGCCAATTGACCATGGTTACGATCGGATCCAGTTACCGGATTTACGCAGGTCAATGCCGTATTGCAC"
    g = orc.Graph.build(65, [seq], 2, True)
    with pytest.raises(aligner.MgxError) as e:
        gpu_graph(g, mode=2)
    assert e.value.code == capi.MGX_ERR_UNSUPPORTED

"""CANONICAL-mode DBGSuccinct graphs (both strands in one graph) on the oracle: groundwork for SURVEY 8(f) rank 1.
dbg_aligner.cpp:225-226 (the reverse-strand seeder always exists), :644-655 (backward pass on the same graph, no RCDBG),
:683-689,:711-722 (alignments on the reverse strand are reported as the forward alignment they mirror), and the generic
branch of Alignment::reverse_complement (alignment.cpp:563-702).  Pinned by the reference's own canonical KATs:
tests/graph/test_aligner.cpp:1483-1537 (CANONICAL half), :1539-1578, and the CLI goldens integration_tests/test_align.py:209-268.
The device path rejects mode != BASIC (MGX_ERR_UNSUPPORTED); the CanonicalDBG wrapper of PRIMARY graphs is not restated yet."""
import os

import orc
from metagraph_amd import capi
from test_oracle_kats import read_fasta, read_fastq, HERE

CANONICAL = 1


def _cfg(**kw):
    c = capi.config_default()
    capi.set_dna_matrix(c, 2, -1, -2)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def test_align_both_directions_canonical():
    # test_aligner.cpp:1539-1578
    k = 7
    reference, query = "AAAAGCTTTCGAGGCCAA", "AAAAGTTTTCGAGGCCAA"
    reference_rc = "TTGGCCTCGAAAGCTTTT"
    g = orc.Graph.build(k, [reference], CANONICAL, True)
    cfg = _cfg()
    run = orc.AlignRun(g, cfg, [query])
    assert run.error == ""
    paths = run.results()[0]
    assert len(paths) == 1
    p = paths[0]
    assert len(p["nodes"]) == 12
    if p["sequence"] == reference:
        assert p["cigar"] == "5=1X12="
    else:
        assert p["sequence"] == reference_rc and p["cigar"] == "12=1X5="
    assert p["score"] == 17 * 2 - 1                      # config.score_sequences(query, reference): 17 matches, C>T is a transition (-1)
    assert p["num_matches"] == 17 and p["clipping"] == 0 and p["end_clipping"] == 0 and p["offset"] == 0
    # check_extend(): MEM-seeded run (max_seed_length = inf) gives the same alignment (test_aligner_helpers.hpp:68-90)
    cfg.max_seed_length = capi.UINT64_MAX
    assert orc.AlignRun(g, cfg, [query]).results()[0] == paths


def test_align_suffix_seed_snp_canonical_mode():
    # test_aligner.cpp:1483-1537, the DeBruijnGraph::CANONICAL iteration
    k = 18
    reference, query = "AAAAACTTTCGAGGCCAA", "GGGGGCTTTCGAGGCCAA"
    reference_rc, query_rc = "TTGGCCTCGAAAGTTTTT", "TTGGCCTCGAAAGCCCCC"
    g = orc.Graph.build(k, [reference_rc], CANONICAL, False)
    cfg = _cfg(max_num_seeds_per_locus=capi.UINT64_MAX, min_cell_score=-2147483648 + 100, min_path_score=-2147483648 + 100,
               min_seed_length=13)
    run = orc.AlignRun(g, cfg, [query])
    assert run.error == ""
    paths = run.results()[0]
    assert len(paths) == 1
    p = paths[0]
    assert len(p["nodes"]) == 1
    if p["sequence"] == reference[5:]:
        assert p["cigar"] == "5S13=" and p["clipping"] == 5 and p["end_clipping"] == 0 and p["score"] == 26
    else:
        assert p["sequence"] == reference_rc[:13]
        assert p["cigar"] == "13=5S" and p["clipping"] == 0 and p["end_clipping"] == 5 and p["score"] == 26
    assert p["offset"] == 5 and p["num_matches"] == 13
    cfg.max_seed_length = capi.UINT64_MAX
    assert orc.AlignRun(g, cfg, [query]).results()[0] == paths


CANONICAL_LINES = {
    0: "MT-10/1\tAACAGAGAATAGTTTAAATTAGAATCTTAGCTTTGGGTGCTAATGGTGGAGTTAAAGACTTTTTCTCTGATTTGTCCTTGGAAAAAGGTTTTCATCTCCGGTTTACAAGACTGGTGTATTAGTTTATACTACAAGGACAGGCCCATTTGA\t+\tAACAGAGAATAGTTTAAATTAGAATCTTAGCTTTGGGTGCTAATGGTGGAGTTAAAGACTTTTTCTCTGATTTGTCCTTGGAAAAAGGTTTTCATCTCCGGTTTACAAGACTGGTGTATTAGTTTATACTACAAGGACAGGCCCATTTGA\t310\t150\t150=\t0",
    1: "MT-8/1\tAAAACTAACCCCCTAATAAAATTAATTAACCACTCATTCATCGACCTCCCCACCCCATCCAACATCTCCGCATGATGAAACTTCGGCTCACTCCTTGGCGCCTGCCTGATCCTCCAAATCACCACAGGACTATTCCTAGCCATGCACTAC\t+\tAAAACTAACCCCCTAATAAAATTAATTAACCACTCATTCATCGACCTCCCCACCCCATCCAACATCTCCGCATGATGAAACTTCGGCTCACTCCTTGGCGCCTGCCTGATCCTCCAAATCACCACAGGACTATTCCTAGCCATGCACTAC\t310\t150\t150=\t0",
    2: "MT-6/1\tATATGACTAGCTTACACAATAGCTTTTATAGTAAAGATACCTCTTTACGGACTCCACTTATGACTCCCTAAAGCCCATGTCGAAGCCCCCATCGCTGGGTCAATAGTACTTGCCGCAGTACTCTTAAAACTAGGCGGCTATGGTATAATA\t+\tATATGACTAGCTTACACAATAGCTTTTATAGTAAAGATACCTCTTTACGGACTCCACTTATGACTCCCTAAAGCCCATGTCGAAGCCCCCATCGCTGGGTCAATAGTACTTGCCGCAGTACTCTTAAAACTAGGCGGCTATGGTATAATA\t310\t150\t150=\t0",
    3: "MT-4/1\tAGTATAGTAGTTCGCTTTGACTGGTGAAGTCTTAGCATGTACTGCTCGGAGGTTCGGTTCTGCTCCGAGGTCGCCCCAACCGAAATTTTTAATGCAGGTTTGGTAGTTTAGGACCTGTGGGTTTGTTAGGTACTGTTTGCATTAATAAAT\t+\tAGTATAGTAGTTCGCTTTGACTGGTGAAGTCTTAGCATGTACTGCTCGGAGGTTGGGTTCTGCTCCGAGGTCGCCCCAACCGAAATTTTTAATGCAGGTTTGGTAGTTTAGGACCTGTGGGTTTGTTAGGTACTGTTTGCATTAATAAAT\t305\t149\t54=1X95=\t0",
    4: "MT-2/1\tTGTGTTAATTAATTAATGCTTGTAGGACATAATAATAACAATTGAATGTCTGCACAGCCACTTTCCACACAGACATCATAACAAAAAATTTCCACCAAACCCCCCCTCCCCCGCTTCTGGCCACAGCACTTAAACACATCTCTGCCAAAC\t+\tTGTGTTAATTAATTAATGCTTGTAGGACATAATAATAACAATTGAATGTCTGCACAGCCACTTTCCACACAGACATCATAACAAAAAATTTCCACCAAACCCCCCCTCCCCCGCTTCTGGCCACAGCACTTAAACACATCTCTGCCAAAC\t310\t150\t150=\t0",
}
SUBK_LINE_5 = ("MT-11/1\tAACAGAGAATTGTTTAAATTACAATCTTAGCTATGGGTGCTAAAGGTGGAGTTATAGACTTTTTCACTGATTTGTCGTTGGAAAAAGCTTTTCATCTCGGGTTTACAAGTCTGGTGTATTTGTTTATACTAGAAGGACAGGCGCATTTGA\t+\t"
               "AACAGAGAATAGTTTAAATTAGAATCTTAGCTTTGGGTGCTAATGGTGGAGTTAAAGACTTTTTCTCTGATTTGTCCTTGGAAAAAGGTTTTCATCTCCGGTTTACAAGACTGGTGTATTAGTTTATACTACAAGGACAGGCCCATTTGA\t245\t137\t"
               "10=1X10=1X10=1X10=1X10=1X10=1X10=1X10=1X10=1X10=1X10=1X10=1X10=1X7=\t0")


def _cli_lines(min_seed_length=None):
    g = orc.Graph.build(11, read_fasta(os.path.join(HERE, "golden", "genome.MT.fa")), CANONICAL, True)
    assert g.num_nodes == 32782
    reads = read_fastq(os.path.join(HERE, "golden", "genome_MT1.fq"))
    cfg = capi.config_cli(11)
    cfg.min_exact_match = 0.0
    if min_seed_length is not None:
        cfg.min_seed_length = min_seed_length
    run = orc.AlignRun(g, cfg, [r[1] for r in reads])
    assert run.error == ""
    lines = run.tsv_lines()
    return [reads[i][0] + l[l.index("\t"):] for i, l in enumerate(lines)]      # the oracle labels lines by index


def test_cli_golden_canonical_mode():
    # integration_tests/test_align.py:209-239 (`metagraph align --align-min-exact-match 0.0` on the canonical graph)
    lines = _cli_lines()
    assert len(lines) == 7
    for i, want in CANONICAL_LINES.items():
        assert lines[i] == want
    last = lines[5].split("\t")
    assert last[0] == "MT-11/1" and last[4] == "22"
    assert last[1] == SUBK_LINE_5.split("\t")[1]


def test_cli_golden_canonical_mode_sub_k_seeds():
    # integration_tests/test_align.py:241-268 (--align-min-seed-length 10)
    lines = _cli_lines(min_seed_length=10)
    assert len(lines) == 7
    for i, want in CANONICAL_LINES.items():
        assert lines[i] == want
    assert lines[5] == SUBK_LINE_5


# integration_tests/test_align.py:124-150: `metagraph align --map --count-kmers` on the canonical genome.MT graph prints
# discovered / k-mers / distinct matched nodes per read (cli/align.cpp:152-164).  That caller uses DeBruijnGraph::map_to_nodes,
# which on a CANONICAL-mode DBGSuccinct takes, per k-mer, the smaller of its own edge and its reverse complement's
# (dbg_succinct.cpp:436-470) — not the aligner's map_to_nodes_sequentially; both strands' sequential mappings give it.
CANONICAL_MAP_COUNTS = ["140/140/140", "140/140/140", "140/140/140", "129/140/129", "140/140/139", "2/140/2", "140/140/140"]


def map_counts(paths):
    return ["%d/%d/%d" % (sum(1 for v in p if v), len(p), len({v for v in p if v})) for p in paths]


def canonical_map_to_nodes(fwd, rev):
    """DBGSuccinct::map_to_nodes in CANONICAL mode from the two sequential mappings (a missing k-mer is missing on both strands)"""
    return [min(a, b) if a and b else 0 for a, b in zip(fwd, rev[::-1])]


def test_cli_map_counts_canonical_mode():
    g = orc.Graph.build(11, read_fasta(os.path.join(HERE, "golden", "genome.MT.fa")), CANONICAL, True)
    reads = read_fastq(os.path.join(HERE, "golden", "genome_MT1.fq"))
    run = orc.AlignRun(g, capi.config_cli(11), [r[1] for r in reads])
    assert map_counts([canonical_map_to_nodes(fwd, rev) for fwd, rev in run.mapping()]) == CANONICAL_MAP_COUNTS

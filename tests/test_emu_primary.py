"""PRIMARY graphs behind the CanonicalDBG wrapper through the kernels' sources (metagraph_amd/csrc/canon_graph.hpp and the
PRIMARY branches of align_core.hpp) under the host model of the wave interface, against the oracle's CanonicalView (pinned by
the wrapper's own KATs, tests/test_oracle_canonical_wrapper.py, and the reference's primary-graph goldens,
tests/test_oracle_primary_goldens.py)."""
import os

import pytest

import emu_drv
import orc
from test_oracle_kats import read_fasta, HERE
from test_oracle_canonical_wrapper import _L as canon_lib, PRIMARY
from test_oracle_primary_goldens import primary_contigs

import ctypes as C


def _orc_children(g, v):
    nodes = (C.c_uint64 * 8)()
    chars = C.create_string_buffer(8)
    n = canon_lib().orc_canonical_adjacent(g.h, v, 0, nodes, chars)
    return [(nodes[i], chars.raw[i:i + 1].decode()) for i in range(n)]


def _graphs():
    mt = read_fasta(os.path.join(HERE, "golden", "genome.MT.fa"))
    yield "mt11-masked", 11, primary_contigs(mt, 11)[0], True
    yield "mt11-unmasked", 11, primary_contigs(mt, 11, "lex")[0], False
    tr = read_fasta(os.path.join(HERE, "golden", "transcripts_100.fa"))[:12]
    yield "tr6-even-k", 6, primary_contigs(tr, 6)[0], True                       # even k: palindromic k-mers
    yield "tr8-even-k-unmasked", 8, primary_contigs(tr, 8, "colex")[0], False
    yield "tr40-k-above-32", 40, primary_contigs(tr[:4], 40)[0], False
    yield "tr64-max-k", 64, primary_contigs(tr[:4], 64, "lex")[0], True
    from test_oracle_canonical_wrapper import DUMMY_GRAPHS                          # test_canonical_dbg.cpp:1239-1640
    for name, seqs in DUMMY_GRAPHS.items():
        yield "dummy-" + name, 31, seqs, False


@pytest.mark.parametrize("name,k,contigs,mask", list(_graphs()), ids=lambda x: x if isinstance(x, str) else None)
def test_wrapper_children_and_terminus_bits(name, k, contigs, mask):
    g = orc.Graph.build(k, contigs, PRIMARY, mask)
    eg = emu_drv.EmuGraph(g, mode=PRIMARY)
    W, last, F, valid = g.export()
    n = g.n_edges
    lib = canon_lib()
    checked = 0
    for u in range(1, n + 1):
        if valid is not None and not valid[u]:
            continue
        for v in (u, u + n):
            want = _orc_children(g, v)
            got, sentinel = eg.canon_children(v)
            assert got == [(x, c) for x, c in want if c != "$"], (name, v)
            deg = lib.orc_canonical_degrees(g.h, v)
            assert eg.terminus_primary(v) == bool((deg & 1) or not (deg & 2)), (name, v)
            checked += 1
    assert checked > (1000 if not name.startswith("dummy-") else 60)


# ---- the whole aligner on PRIMARY graphs: seeds, num_matches and full alignment lists against the oracle ----
import random  # noqa: E402

from metagraph_amd import capi  # noqa: E402
from test_emu_vs_oracle import compare_full, rand_seq, mutate, rc  # noqa: E402
from test_oracle_kats import read_fastq  # noqa: E402
from test_oracle_canonical import _cfg  # noqa: E402


def primary_world(seed, k, genome_len=3000, n_reads=40, read_len=100, mask=False, n_variants=10, order="input"):
    rng = random.Random(seed)
    genome = rand_seq(rng, genome_len)
    seqs = [genome]
    for _ in range(n_variants):
        p = rng.randrange(k, genome_len - k)
        alt = rng.choice([c for c in "ACGT" if c != genome[p]])
        seqs.append(genome[p - k + 1:p] + alt + genome[p + 1:p + k])
    g = orc.Graph.build(k, primary_contigs(seqs, k, order)[0], PRIMARY, mask)
    reads = []
    for i in range(n_reads):
        if i % 10 == 9:
            reads.append(rand_seq(rng, read_len))
            continue
        p = rng.randrange(0, genome_len - read_len)
        r = mutate(rng, genome[p:p + read_len])
        if rng.random() < 0.5:
            r = rc(r)
        if i % 7 == 3:
            r = r[:len(r) // 2] + "N" + r[len(r) // 2 + 1:]
        reads.append(r)
    return g, reads


def test_primary_kats_through_the_kernels():
    # test_aligner.cpp:1483-1537 (PRIMARY iteration) and :1770-1800 (align_suffix_seed_no_full_seeds)
    g = orc.Graph.build(18, ["TTGGCCTCGAAAGTTTTT"], PRIMARY, False)
    cfg = _cfg(max_num_seeds_per_locus=capi.UINT64_MAX, min_cell_score=-2147483648 + 100, min_path_score=-2147483648 + 100,
               min_seed_length=13)
    compare_full(g, emu_drv.EmuGraph(g, mode=PRIMARY), cfg, ["GGGGGCTTTCGAGGCCAA"])
    g = orc.Graph.build(31, ["CTGCTGCGCCATCGCAACCCACGGTTGCTTTTTGAGTCGCTGCTCACGTTAGCCATCACACTGACGTTAAGCTGGCTTTCGATGCTGTATC"],
                        PRIMARY, False)
    query = "CTTACTGCTGCGCTCTTCGCAAACCCCACGGTTTCTTGTTTTGAGCTCGCCTGCTCACGATACCCATACACACTGACGTTCAAGCTGGCTTTCGATGTTGTATC"
    for msl in (0, 131):
        cfg = _cfg(max_num_seeds_per_locus=capi.UINT64_MAX, min_cell_score=-2147483648 + 100,
                   min_path_score=-2147483648 + 100, min_seed_length=13, max_seed_length=msl)
        compare_full(g, emu_drv.EmuGraph(g, mode=PRIMARY), cfg, [query])


@pytest.mark.parametrize("min_seed_length", [None, 10])
@pytest.mark.parametrize("order", ["input", "colex"])
def test_primary_cli_goldens_through_the_kernels(min_seed_length, order):
    contigs, _ = primary_contigs(read_fasta(os.path.join(HERE, "golden", "genome.MT.fa")), 11, order)
    g = orc.Graph.build(11, contigs, PRIMARY, False)
    reads = read_fastq(os.path.join(HERE, "golden", "genome_MT1.fq"))
    cfg = capi.config_cli(11)
    cfg.min_exact_match = 0.0
    if min_seed_length is not None:
        cfg.min_seed_length = min_seed_length
    e = compare_full(g, emu_drv.EmuGraph(g, mode=PRIMARY), cfg, [r[1] for r in reads])
    got, _ = e.results()
    assert (got[3][0]["cigar"], got[3][0]["score"]) == ("54=1X95=", 305)


@pytest.mark.parametrize("k,mask,seed,order", [(11, False, 1, "input"), (19, False, 2, "lex"), (31, False, 3, "colex"),
                                               (15, True, 4, "input"), (12, False, 5, "lex"), (8, True, 6, "input"),
                                               (40, False, 7, "lex"), (63, True, 8, "input"), (64, False, 9, "colex")])
def test_primary_random_worlds(k, mask, seed, order):
    g, reads = primary_world(700 + seed, k, mask=mask, order=order)
    compare_full(g, emu_drv.EmuGraph(g, mode=PRIMARY), capi.config_cli(k), reads)


@pytest.mark.parametrize("k,min_seed,per_locus", [(31, 15, 1000), (31, 13, 2), (21, 9, 1), (12, 6, 1000), (10, 5, 3)])
def test_primary_sub_k_seeding_variants(k, min_seed, per_locus):
    """Sub-k seeds on both strands of a repetitive genome: suffix ranges with several nodes, the per-locus cap, positions that
    collect nodes from the forward phase and from the reverse-complement phase (suffix_to_prefix walks of several levels)."""
    rng = random.Random(78)
    unit = rand_seq(rng, 40)
    genome = rand_seq(rng, 500) + unit + rand_seq(rng, 200) + rc(unit[:30]) + rand_seq(rng, 5) + unit[10:] + rand_seq(rng, 400)
    g = orc.Graph.build(k, primary_contigs([genome], k, "lex")[0], PRIMARY, False)
    reads = []
    for i in range(30):
        p = rng.randrange(0, len(genome) - 120)
        r = mutate(rng, genome[p:p + 120], 0.04)
        reads.append(rc(r) if i % 2 else r)
    cfg = capi.config_cli(k)
    cfg.min_seed_length = min_seed
    cfg.max_num_seeds_per_locus = per_locus
    cfg.min_exact_match = 0.0
    compare_full(g, emu_drv.EmuGraph(g, mode=PRIMARY), cfg, reads)


def test_primary_split_multipass_and_alternatives(monkeypatch):
    monkeypatch.setenv("MGX_EMU_SPLIT", "1")
    monkeypatch.setenv("MGX_EMU_MULTIPASS", "1")
    g, reads = primary_world(750, 15, genome_len=4000, n_reads=40, n_variants=40)
    chim = [reads[i][:50] + reads[i + 1][40:90] for i in range(0, 20, 2)]
    eg = emu_drv.EmuGraph(g, mode=PRIMARY)
    for n_alt, msl in ((1, 15), (2, 11)):
        cfg = capi.config_cli(15)
        cfg.min_exact_match = 0.0
        cfg.num_alternative_paths = n_alt
        cfg.min_seed_length = msl
        compare_full(g, eg, cfg, reads + chim)


def test_primary_forward_only_and_general_path(monkeypatch):
    g, reads = primary_world(760, 13, n_reads=30, mask=True)
    eg = emu_drv.EmuGraph(g, mode=PRIMARY)
    cfg = capi.config_cli(13)
    cfg.forward_and_reverse_complement = 0            # a CanonicalDBG aligns both strands regardless (dbg_aligner.cpp:225-226)
    compare_full(g, eg, cfg, reads)
    monkeypatch.setenv("MGX_NO_FAST", "1")            # every column through general_step
    compare_full(g, eg, capi.config_cli(13), reads)


def test_primary_mapping_is_the_wrappers():
    g, reads = primary_world(770, 12, n_reads=20)
    eg = emu_drv.EmuGraph(g, mode=PRIMARY)
    lib = canon_lib()
    e = emu_drv.EmuRun(eg, capi.config_cli(12), reads, map_only=True)
    for q, ((fwd, rev), r) in enumerate(zip(e.mapping(), reads)):
        out = (C.c_uint64 * (len(r) - 12 + 1))()
        lib.orc_canonical_map(g.h, r.encode(), len(r), out)
        assert list(fwd) == list(out), q
        mirrored = [int(lib.orc_canonical_reverse_complement(g.h, v)) if v else 0 for v in out][::-1]
        assert list(rev) == mirrored, q


@pytest.mark.parametrize("lanes,tables", [("8", "1"), ("", "0"), ("8", "0")])
def test_primary_in_the_8_lane_group_model_and_without_tables(lanes, tables):
    """The same checks with 8 lanes per read (the extension kernel's geometry), and with MGX_PRIMARY_TABLES=0: the wrapper
    re-deriving spellings and look-ups per expansion instead of reading the reverse-complement tables (a load-time choice, so
    these run in subprocesses)."""
    import subprocess
    import sys
    env = dict(os.environ, MGX_EMU_WAVE=lanes, MGX_PRIMARY_TABLES=tables)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", os.path.abspath(__file__), "-k",
                        "random_worlds or split_multipass or sub_k_seeding or wrapper_children or cli_goldens or kats",
                        "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_align_low_similarity4_rep_primary_through_the_kernels():
    # tests/graph/test_aligner.cpp:1602-1634: three alignments of a low-complexity query on the k = 6 primary transcripts graph
    from test_alt_paths import QUERY
    contigs, _ = primary_contigs(read_fasta(os.path.join(HERE, "golden", "transcripts_100.fa")), 6)
    g = orc.Graph.build(6, contigs, PRIMARY, True)
    c = capi.config_default()
    capi.set_dna_matrix(c, 2, -3, -3)
    c.gap_opening_penalty, c.gap_extension_penalty = -5, -2
    c.xdrop = 27
    c.min_exact_match = 0.0
    c.max_nodes_per_seq_char = 10.0
    c.num_alternative_paths = 3
    c.min_seed_length = 6
    lim = capi.Limits()
    lim.max_columns = 60000
    want = orc.AlignRun(g, c, [QUERY]).results()
    assert len(want[0]) == 3
    e = emu_drv.EmuRun(emu_drv.EmuGraph(g, mode=PRIMARY), c, [QUERY], limits=lim)
    assert e.error == ""
    got, status = e.results()
    assert status == [0] and got == want


def test_cli_map_counts_through_the_wrapper_and_the_kernels():
    """`metagraph align --map --count-kmers` wraps a PRIMARY graph into CanonicalDBG first (cli/align.cpp:345-348): every k-mer
    of the canonical graph is found through the wrapper, so discovered / k-mers are those of the canonical-graph golden
    (integration_tests/test_align.py:124-150); the device mapping path returns the wrapper's ids."""
    from test_oracle_canonical import CANONICAL_MAP_COUNTS
    contigs, _ = primary_contigs(read_fasta(os.path.join(HERE, "golden", "genome.MT.fa")), 11, "lex")
    g = orc.Graph.build(11, contigs, PRIMARY, True)
    reads = [r[1] for r in read_fastq(os.path.join(HERE, "golden", "genome_MT1.fq"))]
    lib = canon_lib()
    paths = []
    for r in reads:
        out = (C.c_uint64 * (len(r) - 11 + 1))()
        lib.orc_canonical_map(g.h, r.encode(), len(r), out)
        paths.append(list(out))
    assert ["%d/%d" % (sum(1 for v in p if v), len(p)) for p in paths] == [c.rsplit("/", 1)[0] for c in CANONICAL_MAP_COUNTS]
    e = emu_drv.EmuRun(emu_drv.EmuGraph(g, mode=PRIMARY), capi.config_cli(11), reads, map_only=True)
    assert [list(fwd) for fwd, _ in e.mapping()] == paths

"""The oracle's aligner on PRIMARY graphs behind the CanonicalDBG wrapper against the reference's own goldens for that mode:
`metagraph align` on the k = 11 primary genome.MT graph (integration_tests/test_align.py:271-299 all lines, :301-329 with
--align-min-seed-length 10) and align_low_similarity4_rep_primary (tests/graph/test_aligner.cpp:1602-1634).

A PRIMARY graph holds one k-mer of every {k-mer, reverse complement} pair.  The reference extracts them as "primary contigs"
(`metagraph build --mode canonical` + `transform --to-fasta --primary-kmers`, graph/representation/succinct/boss.cpp
call_sequences(kmers_in_single_form = true)); which member of a pair ends up stored depends on that traversal's order, which is
not restated here.  `primary_contigs` below is a test-side path cover with the same contract (every pair exactly once), and the
goldens are asserted for three different visiting orders: the aligner's output must not depend on the choice."""
import os

import pytest

import orc
from metagraph_amd import capi
from test_oracle_kats import read_fasta, read_fastq, HERE
from test_oracle_canonical import CANONICAL_LINES, SUBK_LINE_5
from test_alt_paths import QUERY

PRIMARY = 2
_COMP = str.maketrans("ACGT", "TGCA")


def rc(s):
    return s.translate(_COMP)[::-1]


def primary_contigs(seqs, k, order="input"):
    kmers, seen_order = set(), []
    for s in seqs:
        for t in (s, rc(s)):
            for i in range(len(t) - k + 1):
                km = t[i:i + k]
                if km not in kmers and all(c in "ACGT" for c in km):
                    kmers.add(km)
                    seen_order.append(km)
    if order == "lex":
        seen_order = sorted(kmers)
    elif order == "colex":
        seen_order = sorted(kmers, key=lambda s: s[::-1])
    visited, out = set(), []
    for km in seen_order:
        if km in visited:
            continue
        visited.update((km, rc(km)))
        path, cur = km, km
        while True:
            for c in "ACGT":
                nxt = cur[1:] + c
                if nxt in kmers and nxt not in visited:
                    break
            else:
                break
            visited.update((nxt, rc(nxt)))
            path += c
            cur = nxt
        out.append(path)
    return out, len(kmers)


@pytest.fixture(scope="module", params=["input", "lex", "colex"])
def mt_primary(request):
    contigs, n_both = primary_contigs(read_fasta(os.path.join(HERE, "golden", "genome.MT.fa")), 11, request.param)
    assert n_both == 32782                                   # the canonical graph's `nodes (k)` (test_align.py:216)
    masked = orc.Graph.build(11, contigs, PRIMARY, True)
    assert masked.num_nodes == 16391                         # test_align.py:280: 'nodes (k)' of the primary graph
    return orc.Graph.build(11, contigs, PRIMARY, False)      # `align` drops the dummy mask (cli/align.cpp)


def _lines(g, min_seed_length=None):
    reads = read_fastq(os.path.join(HERE, "golden", "genome_MT1.fq"))
    cfg = capi.config_cli(11)
    cfg.min_exact_match = 0.0
    if min_seed_length is not None:
        cfg.min_seed_length = min_seed_length
    run = orc.AlignRun(g, cfg, [r[1] for r in reads])
    assert run.error == ""
    return [reads[i][0] + l[l.index("\t"):] for i, l in enumerate(run.tsv_lines())]


def test_cli_golden_primary(mt_primary):
    lines = _lines(mt_primary)
    assert len(lines) == 7
    for i, want in CANONICAL_LINES.items():                   # test_align.py:291-295 == the canonical-mode lines
        assert lines[i] == want
    assert lines[6].split("\t")[4] == "310"
    last = lines[5].split("\t")
    assert last[0] == "MT-11/1" and last[4] == "22"
    assert last[1] == SUBK_LINE_5.split("\t")[1]


def test_cli_golden_primary_sub_k_seeds(mt_primary):
    lines = _lines(mt_primary, min_seed_length=10)
    assert len(lines) == 7
    for i, want in CANONICAL_LINES.items():
        assert lines[i] == want
    assert lines[5] == SUBK_LINE_5                           # test_align.py:328


@pytest.mark.parametrize("order", ["input", "lex"])
def test_align_low_similarity4_rep_primary(order):
    contigs, _ = primary_contigs(read_fasta(os.path.join(HERE, "golden", "transcripts_100.fa")), 6, order)
    g = orc.Graph.build(6, contigs, PRIMARY, True)
    c = capi.config_default()
    capi.set_dna_matrix(c, 2, -3, -3)
    c.gap_opening_penalty, c.gap_extension_penalty = -5, -2
    c.xdrop = 27
    c.min_exact_match = 0.0
    c.max_nodes_per_seq_char = 10.0
    c.num_alternative_paths = 3
    c.min_seed_length = 6
    for _ in range(3):
        (paths,) = orc.AlignRun(g, c, [QUERY]).results()
        assert len(paths) == 3

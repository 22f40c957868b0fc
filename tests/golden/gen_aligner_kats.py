#!/usr/bin/env python3
"""Known-answer vectors for the aligner, transcribed BY HAND from the reference's unit tests
(metagraph/tests/graph/test_aligner.cpp, the DBGSuccinct instantiation of each TYPED_TEST) and
from the CLI goldens in metagraph/integration_tests/test_align.py.

This file holds DATA only: graph sequences, k, config overrides, queries and the values the
reference's tests assert.  Expected scores that the reference writes as expressions
(`config.match_score(query) + config.gap_opening_penalty`) are evaluated here with the score
matrix the test configures.  Run to (re)write aligner_kats.json:  python gen_aligner_kats.py
"""
import json
import os

INT_MIN = -2**31
NINF = INT_MIN + 100


def dna_matrix(match, transition, transversion):
    # DBGAlignerConfig::dna_scoring_matrix (aligner_config.cpp:164-183)
    def score(a, b):
        if a == b and a in "ACGT":
            return match
        if {a, b} in ({"A", "G"}, {"C", "T"}):
            return transition
        return transversion
    return score


def unit_matrix(match):
    # DBGAlignerConfig::unit_scoring_matrix over "ACGT" (aligner_config.cpp:185-204)
    def score(a, b):
        return match if (a == b and a in "ACGT") else -match
    return score


def mk_score(matrix):
    return dna_matrix(*matrix[1:]) if matrix[0] == "dna" else unit_matrix(matrix[1])


def ss(matrix, a, b):
    f = mk_score(matrix)
    assert len(a) == len(b)
    return sum(f(x, y) for x, y in zip(a, b))


def ms(matrix, q):
    return ss(matrix, q, q)


def rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


D212 = ["dna", 2, -1, -2]
D233 = ["dna", 2, -3, -3]
D211 = ["dna", 2, -1, -1]
CASES = []


def case(name, line, k, seqs, query, matrix=D212, cfg=None, expect=None, mask_dummy=True,
         check_extend=True, note=None):
    c = {"name": name, "ref": "tests/graph/test_aligner.cpp:%d" % line, "k": k, "graph": seqs,
         "mask_dummy": mask_dummy, "query": query, "matrix": matrix, "config": cfg or {},
         "expect": expect or {}, "check_extend": check_extend}
    if note:
        c["note"] = note
    CASES.append(c)


def exp(n_nodes, sequence, score, cigar, num_matches, clipping=0, end_clipping=0, offset=0, **kw):
    e = {"n_paths": 1, "n_nodes": n_nodes, "sequence": sequence, "score": score,
         "num_matches": num_matches, "clipping": clipping, "end_clipping": end_clipping, "offset": offset}
    if isinstance(cigar, list):
        e["cigar_any"] = cigar
    else:
        e["cigar"] = cigar
    e.update(kw)
    return e


case("bad_min_cell_score", 96, 3, [], "", cfg={"min_cell_score": INT_MIN, "min_path_score": INT_MIN},
     expect={"throws": True}, check_extend=False)
case("align_empty", 105, 4, ["CATTT"], "", expect={"n_paths": 0}, check_extend=False)
case("align_sequence_much_too_short", 119, 4, ["CATTT"], "CA", expect={"n_paths": 0}, check_extend=False)
case("align_sequence_too_short", 133, 4, ["CATTT"], "CAT", cfg={"min_seed_length": 4},
     expect={"n_paths": 0}, check_extend=False)
q = "AAAAAAAAA"
case("align_big_self_loop", 148, 3, ["AAAA"], q, expect=exp(7, q, ms(D212, q), "9=", 9))
case("align_single_node", 176, 3, ["CAT"], "CAT", expect=exp(1, "CAT", ms(D212, "CAT"), "3=", 3))
r = "AGCTTCGAGGCCAA"
case("align_straight", 204, 4, [r], r, expect=exp(11, r, ms(D212, r), "14=", 14))
case("align_straight_min_path_score", 248, 4, [r], r, cfg={"min_path_score": 100}, expect={"n_paths": 0})
q = "AGCTNCGAGGCCAA"
case("align_straight_with_N", 264, 4, [r], q, expect=exp(11, r, ss(D212, r, q), "4=1X9=", 13))
case("align_straight_forward_and_reverse_complement", 296, 4, [r], rc(r),
     expect=exp(11, r, ms(D212, rc(r)), "14=", 14, orientation=1))
# the same through align_batch (:338-385); align_straight_max_size (:225-246) is commented out upstream
case("align_straight_forward_and_reverse_complement_batch", 338, 4, [r], rc(r),
     expect=exp(11, r, ms(D212, rc(r)), "14=", 14, orientation=1))
r1, r2 = "AGCTTCGAA", "AGCTTCGAC"
case("align_ending_branch", 387, 4, [r1, r2], r2, expect=exp(6, r2, ms(D212, r2), "9=", 9))
r1, r2 = "AGCTTCGAATATTTGTT", "AGCTTCGACGATTTGTT"
case("align_branch", 418, 6, [r1, r2], r2, expect=exp(12, r2, ms(D212, r2), "17=", 17))
case("align_branch_with_cycle", 449, 4, [r1, r2], r2, expect=exp(14, r2, ms(D212, r2), "17=", 17))
q = "AGGGGG"
case("repetitive_sequence_alignment", 480, 3, ["AGGGGGGGGGAAAAGGGGGGG"], q, expect=exp(4, q, ms(D212, q), "6=", 6))
r, q = "AGCAACTCGAAA", "AGCAATTCGAAA"
case("variation", 508, 4, [r], q, expect=exp(9, r, ss(D212, q, r), "5=1X6=", 11))
r1, r2, q = "TTAAGCAACTCGAAA", "TTAAGCAAGTCGAAA", "TTAAGCAATGGGAAA"
case("variation_in_branching_point", 538, 4, [r1, r2], q, cfg={"gap_opening_penalty": -3, "gap_extension_penalty": -1},
     expect={"n_paths": 1, "n_nodes": 12, "sequence_any": [r1, r2], "cigar": "8=3X4=", "num_matches": 12,
             "clipping": 0, "end_clipping": 0, "offset": 0})
r, q = "ACGCAACTCTCTGAACTTGT", "ACGCAATTCTCTGTATTTGT"
case("multiple_variations", 578, 4, [r], q, expect=exp(17, r, ss(D212, q, r), "6=1X6=1X1=1X4=", 17))
case("align_noise_in_branching_point", 608, 4, ["AAAACTTTTTT", "AAAATTGGGGG"], "AAAATTTTTTT", matrix=D233,
     cfg={"gap_opening_penalty": -3, "gap_extension_penalty": -1},
     expect={"n_paths": 1, "n_nodes": 9, "num_matches": 11, "clipping": 0, "end_clipping": 0, "offset": 0,
             "by_orientation": {"0": {"sequence": "AAAACTTTTTTT", "cigar": "4=1D7="},
                                "1": {"sequence": "AAAAAAACTTTT", "cigar": "7=1D4="}}})
case("alternative_path_basic", 650, 4, ["ACAATTTTTTTT", "ACAATTTTTGTT", "ACAAGTTTTTTT", "ACAAGTTTTGTT"],
     "ACAACTTTTCTT", cfg={"gap_opening_penalty": -3, "gap_extension_penalty": -1, "num_alternative_paths": 2},
     expect={"n_paths": 2, "first": {"cigar": "4=1X4=1X2=", "num_matches": 10, "clipping": 0, "end_clipping": 0,
                                     "offset": 0}},
     note="num_alternative_paths > 1: tie order among equal alignments is unpinned (SURVEY 8c)")
r, q = "AAAGCGGACCCTTTCCGTTAT", "AAAGGGGACCCTTTTCGTTAT"
case("align_multiple_misalignment", 682, 4, [r], q, expect=exp(18, r, ss(D212, q, r), "4=1X9=1X6=", 19))
r = "TTTCCTTGTT"
g33 = {"gap_opening_penalty": -3, "gap_extension_penalty": -3}
case("align_insert_non_existent", 712, 4, [r], "TTTCCATTGTT", cfg=g33, expect=exp(7, r, ms(D212, r) - 3, "5=1I5=", 10))
case("align_insert_multi", 744, 4, [r], "TTTCCAATTGTT", cfg=g33, expect=exp(7, r, ms(D212, r) - 3 - 3, "5=2I5=", 10))
g11 = {"gap_opening_penalty": -1, "gap_extension_penalty": -1}
case("align_insert_long", 777, 4, [r], "TTTCCAAAAAAAAATTGTT", matrix=D211, cfg=g11,
     expect=exp(7, r, ms(D211, r) - 1 - 8, "5=9I5=", 10))
r = "TTTCCGGTTGTTA"
case("align_insert_long_offset", 810, 5, [r], "TTTCCGCAAAAAAAAATTGTTA", matrix=D211, cfg=g11,
     expect=exp(9, r, ss(D211, r, "TTTCCGCTTGTTA") - 1 - 8, ["6=1X9I6=", "6=9I1X6="], 12))
r, q = "TTCGATTGGCCT", "TTCGATGGCCT"
case("align_delete", 845, 4, [r], q, cfg=g33, expect=exp(9, r, ms(D212, q) - 3, ["6=1D5=", "5=1D6="], None),
     check_extend=False)
r, q = "TTTCTGTATACCTTGGCGCTCTC", "TTTCTGTATAGGCGCTCTC"
case("align_gap", 883, 4, [r], q, cfg=g33, expect=exp(20, r, ms(D212, q) - 3 - 9, "10=4D9=", 19))
r, q = "TTTCCCTTGGCGCTCTC", "TTTCGGCGCTCTC"
case("align_gap_after_seed", 916, 4, [r], q, cfg={"gap_opening_penalty": -3, "gap_extension_penalty": -1},
     expect=exp(14, r, ms(D212, q) - 3 - 3, "4=4D9=", 13))
q = "AAAACGAGGCCAA"
U1 = ["unit", 1]
case("align_loop_deletion", 949, 4, ["AAAATTTTCGAGGCCAA"], q, matrix=U1, cfg=g11,
     expect=exp(13, "AAAATTTCGAGGCCAA", ms(U1, q) - 1 - 2, "4=3D9=", 13))
r1 = "AGCTTCGAGGCCAAGCCTGACTGATCGATGCATGCTAGCTAGTCAGTCAGCGTGAGCTAGCAT"
r2 = "AGCTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT"
case("align_straight_long_xdrop", 983, 4, [r1, r2], r1, matrix=D233, cfg={"xdrop": 30, "rel_score_cutoff": 0.8},
     expect=exp(60, r1, ms(D233, r1), "63=", 63))
r, q = "TTTCCCTGGCGCTCTC", "TTTCCGGGGCGCTCTC"
D266 = ["dna", 2, -6, -6]
case("align_drop_seed", 1016, 4, [r], q, matrix=D266,
     cfg={"gap_opening_penalty": -10, "gap_extension_penalty": -4, "xdrop": 6},
     expect=exp(6, r[7:], ms(D266, r[7:]), "7S9=", 9, clipping=7))
r, q = "TTTCCCTTAAGGCGCTCTC", "TTTCGGCGCTCTC"
case("align_long_gap_after_seed", 1052, 4, [r], q, cfg={"gap_opening_penalty": -5, "gap_extension_penalty": -1},
     expect=exp(6, r[10:], ms(D212, q[4:]), "4S9=", 9, clipping=4))
r = "TTTGTGGCTAGAGCTCGAGATCGCGCGGCCACAATTGACAAATGAGATCTAATTAAACTAAAGAGCTTCTGCACAGCAAAAGAAACTGTCATC"
q = "TTTGTGGCTAGAGCTCGAGATCGCGCGGCCACAATTGACAAATGACAAATGTGATCTAATGAAACTAAAGAGCTTCTGCACAGCAAAAGAAACTGTCATC"
b = "TTTGTGGCTAGAGCTCGAGATCGCGCGGCCACAATTGACAAATGAGATCTAATGAAACTAAAGAGCTTCTGCACAGCAAAAGAAACTGTCATC"
case("align_repeat_sequence_no_delete_after_insert", 1085, 27, [r], q, matrix=D233, cfg=g33,
     expect=exp(67, r, ss(D233, r, b) - 3 - 18,
                ["45=7I8=1X39=", "45=5I1=2I7=1X39=", "44=2I1=5I8=1X39=", "44=3I1=4I8=1X39=", "44=4I1=3I8=1X39="], 92),
     check_extend=False, note="the extended (max_seed_length=inf) run must satisfy the same expectations")
CASES[-1]["extend_same_expect"] = True
r, q = "GGCCTGTTTG", "ACCCTGTTTG"
case("align_clipping1", 1149, 4, [r], q, expect=exp(5, r[2:], ms(D212, q[2:]), "2S8=", 8, clipping=2))
r, q = "AAAAGCTTCGAGGCCAA", "TTAGCTTCGAGGCCAA"
case("align_clipping2", 1179, 4, [r], q, expect=exp(11, r[3:], ms(D212, q[2:]), "2S14=", 14, clipping=2))
r, q = "TTTTTTTAAAAGCTTCGAGGCCAA", "CCCCCCCAAAAGCTTCGAGGCCAA"
case("align_long_clipping", 1208, 4, [r], q, expect=exp(14, r[7:], ms(D212, q[7:]), "7S17=", 17, clipping=7))
r, q = "AAAAGCTTCGAGGCCAATTTTTTT", "AAAAGCTTCGAGGCCAACCCCCCC"
case("align_end_clipping", 1239, 4, [r], q, expect=exp(14, r[:17], ms(D212, q[:17]), "17=7S", 17, end_clipping=7))
r, q = "AAAAGCTTTCGAGGCCAA", "ACCTTTCGAGGCCAA"
case("align_clipping_min_cell_score", 1268, 7, [r], q, cfg={"min_cell_score": NINF, "min_path_score": NINF},
     expect=exp(7, r[5:], ms(D212, q[2:]), "2S13=", 13, clipping=2))
case("align_low_similarity", 1299, 27, ["CTAGAACTTAAAGTATAATAATACTAATAATAAAATAAAATACA"],
     "CTAGAACTTAAAGTATAATAATACTAATAAAAGTACAATACA", matrix=D233, expect={"n_paths": 1}, check_extend=False)
case("align_low_similarity2", 1330, 27, ["GCCACAATTGACAAATGAGATCTAATTAAACTAAAGAGCTTCTGCACAGCAAAAGAAACTGTCATC"],
     "GCCACAATTGACAAATGACAAATGTGATCTAATGAAACTAAAGAGCTTCTGCACAGCAAAAGAAACTGTCATC", matrix=D233,
     expect={"n_paths": 1}, check_extend=False)
r = "AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAGTGCTGGGATTATAGGTGTGAACCACCACACCTGGCTAATTTTTTTTGTGTGTGTGTGTGTTTTTTC"
q = "AAAAAAAAAAAAAAAAAAAAAAAAAAACGCCAAAAAGGGGGAATAGGGGGGGGGGAACCCCAACACCGGTATGTTTTTTTGTGTGTGGGGGATTTTTTTC"
case("align_low_similarity3_nofilter", 1345, 27, [r], q, matrix=D233, cfg={"seed_complexity_filter": 0},
     expect={"n_paths_min": 1}, check_extend=False)
case("align_low_similarity3_filter", 1345, 27, [r], q, matrix=D233, cfg={"seed_complexity_filter": 1},
     expect={"n_paths": 0}, check_extend=False, note="the only reference test that pins sdust")
r = "GTCGTCAGATCGGAAGAGCGTCGTGTAGGGAAAGGTCTTCGCCTGTGTAGATCTCGGTGGTCG"
q = "GTCAGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGTTCCTGGTGGTGTAGATC"
case("align_low_similarity5", 1426, 31, [r], q, matrix=D233, expect={"n_paths": 1}, mask_dummy=False)
r, q = "AAAAGCTTTCGAGGCCAA", "ACCTTTCGAGGCCAA"
case("align_suffix_seed_snp_min_seed_length", 1447, 7, [r], q, mask_dummy=False,
     cfg={"min_seed_length": 2, "max_num_seeds_per_locus": 2**64 - 1, "min_cell_score": NINF, "min_path_score": NINF},
     expect=exp(7, r[5:], ms(D212, q[2:]), "2S13=", 13, clipping=2))
r, q = "GTAGTGCTAGCTGTAGTCGTGCTGATGC", "GTAGTGCTACCTGTAGTCGTGGTGATGC"
case("align_both_directions2", 1571, 11, [r], q, expect={"n_paths": 1, "n_nodes": 18, "sequence": r,
                                                          "score": ss(D212, q, r)})
r, q = "AAAAGCTTTCGAGGCCAA", "AAAAGTTTTCGAGGCCAA"
case("align_nodummy_fwd_only", 1627, 7, [r], q, cfg={"forward_and_reverse_complement": 0},
     expect=exp(6, r[6:], ss(D212, q[6:], r[6:]), "6S12=", 12, clipping=6))
case("align_nodummy_both", 1627, 7, [r], q, cfg={"forward_and_reverse_complement": 1},
     expect=exp(12, r, ss(D212, q, r), "5=1X12=", 17))
case("align_seed_to_end", 1675, 5, ["ATCCCTTTTAAAA"], "ATCCCGGGGGGGGGGGGGGGGGTTTTAAAA", expect={"n_paths": 1})
r1 = "TCGGGGCAAGAAACACACAGCCTTCTCATCCAAGGGCCTCAGTGATGAAGAGTACGATGAGTACAAGAGGATCAGAGAAGAAAGGAATGGCAAATACTCCATAGAAGAGTACCTTCAGGACAGGGACAGATACTATGAGGAGGTGGCCAT"
r2 = "TCGGGGCAAGAAACACACAGCCTTCTCATCCAAGGGCCTCAGTGATGAAGAGTACGATGAGTACAAGAGAATCAGAGAGGAGAGGAATGGCAAATACTCAATAGAGGAATACCTCCAAGATAGGGACAGATACTATGAAGAGCTTGCCAT"
q = "TCGGGGCAAGAAACACACAGCCTTCTCATCCAAGGGCCTCAGTGATGATGAGTACGATGAGTACAAGAGCATCAGAGAGGAGAGGAATGGCAAATACTCAATAGAGGAATACCTCCAAGATAGGGACAGATACTATGAAGAGCTTGCCAT"
case("align_bfs_vs_dfs_xdrop", 1692, 31, [r1, r2], q, matrix=D233,
     cfg={"xdrop": 27, "min_seed_length": 0, "max_seed_length": 0, "rel_score_cutoff": 0.8},
     expect={"n_paths": 1, "cigar": "48=1X20=1X80="}, check_extend=False)
r, q = "AAAAGCTTTCGAGGCCAA", "AAAAGTTTTCGAGGCCAA"
case("align_dummy", 1715, 7, [r], q, mask_dummy=False, cfg={"min_seed_length": 5},
     expect=exp(12, r, ss(D212, q, r), "5=1X12=", 17))
r1 = "CGTGGCCCAGGCCCAGGCCCAGGCCCAGGCCCAGGCCCAGGCCCAGGCCCAGGCCCAGGCCCAGGCCCAAGCC"
r2 = "CGTGGCCCAGGCCCAGGCCCAGCCCCAGGCCCAGGCCCAGGCCCAGGCCCAGGCCCAGGCCCAGGCCCAAGCC"
q = "CGTGGCCCAGGCCCAGGCCCAGTGGGCGTTGGCCCAGGCGGCCACGGTGGCTGCGCAGGCCCGCCTGGCACAAGCCACGCTG"
case("align_extended_insert_after_match", 1746, 27, [r1, r2], q, matrix=D233, mask_dummy=False,
     cfg={"min_seed_length": 15}, expect={"n_paths": 1, "score": 52})

# ---------------------------------------------------------------------------------------------------
# CLI goldens: integration_tests/test_align.py (graph: genome.MT.fa, k=11, BASIC succinct; reads:
# genome_MT1.fq; `metagraph align` defaults with --align-min-exact-match 0.0).  Whole TSV lines.
# ---------------------------------------------------------------------------------------------------
MT10 = "AACAGAGAATAGTTTAAATTAGAATCTTAGCTTTGGGTGCTAATGGTGGAGTTAAAGACTTTTTCTCTGATTTGTCCTTGGAAAAAGGTTTTCATCTCCGGTTTACAAGACTGGTGTATTAGTTTATACTACAAGGACAGGCCCATTTGA"
MT8 = "AAAACTAACCCCCTAATAAAATTAATTAACCACTCATTCATCGACCTCCCCACCCCATCCAACATCTCCGCATGATGAAACTTCGGCTCACTCCTTGGCGCCTGCCTGATCCTCCAAATCACCACAGGACTATTCCTAGCCATGCACTAC"
MT6 = "ATATGACTAGCTTACACAATAGCTTTTATAGTAAAGATACCTCTTTACGGACTCCACTTATGACTCCCTAAAGCCCATGTCGAAGCCCCCATCGCTGGGTCAATAGTACTTGCCGCAGTACTCTTAAAACTAGGCGGCTATGGTATAATA"
MT4 = "AGTATAGTAGTTCGCTTTGACTGGTGAAGTCTTAGCATGTACTGCTCGGAGGTTCGGTTCTGCTCCGAGGTCGCCCCAACCGAAATTTTTAATGCAGGTTTGGTAGTTTAGGACCTGTGGGTTTGTTAGGTACTGTTTGCATTAATAAAT"
MT2 = "TGTGTTAATTAATTAATGCTTGTAGGACATAATAATAACAATTGAATGTCTGCACAGCCACTTTCCACACAGACATCATAACAAAAAATTTCCACCAAACCCCCCCTCCCCCGCTTCTGGCCACAGCACTTAAACACATCTCTGCCAAAC"
MT11 = "AACAGAGAATTGTTTAAATTACAATCTTAGCTATGGGTGCTAAAGGTGGAGTTATAGACTTTTTCACTGATTTGTCGTTGGAAAAAGCTTTTCATCTCGGGTTTACAAGTCTGGTGTATTTGTTTATACTAGAAGGACAGGCGCATTTGA"

CLI = {
    "graph_fasta": "genome.MT.fa", "reads_fastq": "genome_MT1.fq", "k": 11,
    "expected_num_real_kmers": 16438,     # test_align.py:33 'nodes (k)' with --mask-dummy
    "runs": [
        {"name": "test_simple_align_all_graphs", "ref": "integration_tests/test_align.py:26-57",
         "flags": {"forward_and_reverse_complement": 0, "min_exact_match": 0.0},
         "lines": {
             "0": "MT-10/1\t" + MT10 + "\t+\tTAGAATCTTAG\t22\t11\t19S11=120S\t0",
             "1": "MT-8/1\t" + MT8 + "\t+\t" + MT8 + "\t310\t150\t150=\t0",
             "2": "MT-6/1\t" + MT6 + "\t+\t" + MT6 + "\t310\t150\t150=\t0",
             "3": "MT-4/1\t" + MT4 + "\t*\t*\t0\t*\t*\t*",
             "4": "MT-2/1\t" + MT2 + "\t+\t" + MT2 + "\t310\t150\t150=\t0"},
         "fields": {"5": {"0": "MT-11/1", "1": MT11, "4": "22"}}, "n_lines": 7},
        {"name": "test_simple_align_fwd_rev_comp_all_graphs", "ref": "integration_tests/test_align.py:175-206",
         "flags": {"forward_and_reverse_complement": 1, "min_exact_match": 0.0},
         "lines": {
             "0": "MT-10/1\t" + MT10 + "\t-\t" + rc(MT10) + "\t310\t150\t150=\t0",
             "1": "MT-8/1\t" + MT8 + "\t+\t" + MT8 + "\t310\t150\t150=\t0",
             "2": "MT-6/1\t" + MT6 + "\t+\t" + MT6 + "\t310\t150\t150=\t0",
             "3": "MT-4/1\t" + MT4 + "\t-\tATTTATTAATGCAAACAGTACCTAACAAACCCACAGGTCCTAAACTACCAAACCTGCATTAAAAATTTCGGTTGGGGCGACCTCGGAGCAGAACCCAACCTCCGAGCAGTACATGCTAAGACTTCACCAGTCAAAGCGAACTACTATACT\t305\t149\t95=1X54=\t0",
             "4": "MT-2/1\t" + MT2 + "\t+\t" + MT2 + "\t310\t150\t150=\t0"},
         "fields": {"5": {"0": "MT-11/1", "1": MT11, "4": "22"}}, "n_lines": 7},
    ],
    # `align --map --count-kmers` (test_align.py:59-87): matched/total/unique k-mers per read
    "map_counts": {"ref": "integration_tests/test_align.py:59-87",
                   "counts": ["1/140/1", "140/140/140", "140/140/140", "0/140/0", "140/140/140", "1/140/1", "1/140/1"]},
}

if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "aligner_kats.json"), "w") as f:
        json.dump({"unit": CASES, "cli": CLI}, f, indent=1)
    print("wrote %d unit cases" % len(CASES))

"""The per-query digest used by the oracle-free property tests (tests/result_digest.py), checked on the oracle's own results:
permuting a batch permutes the digests, and a change anywhere in an alignment changes that query's digest."""
import ctypes as C
import random

import numpy as np

import orc
from metagraph_amd import capi
from result_digest import query_digests
from test_emu_vs_oracle import make_world


def _oracle_results(g, cfg, reads):
    run = orc.AlignRun(g, cfg, reads)
    res = capi.Results()
    orc.L().orc_results_view(run.r, C.byref(res))
    return run, res


def test_digests_follow_a_permutation_and_see_every_field():
    g, reads = make_world(4242, 15, n_reads=60)
    cfg = capi.config_cli(15)
    cfg.num_alternative_paths = 2
    cfg.min_exact_match = 0.0
    run1, res1 = _oracle_results(g, cfg, reads)
    d1 = query_digests(res1)
    assert len(d1) == len(reads) and len(set(d1.tolist())) > len(reads) // 2
    perm = list(range(len(reads)))
    random.Random(1).shuffle(perm)
    run2, res2 = _oracle_results(g, cfg, [reads[p] for p in perm])
    assert np.array_equal(query_digests(res2), d1[perm])
    # sensitivity: flip one value of each kind in place (the views alias the oracle's buffers) and look at the digests again
    a = capi.results_arrays(res1)
    q = next(i for i in range(len(reads)) if a["aln_begin"][i + 1] > a["aln_begin"][i])
    first = int(a["aln_begin"][q])
    for arr, idx in ((a["nodes"], int(a["alns"]["nodes_begin"][first])), (a["seqs"], int(a["alns"]["seq_begin"][first])),
                     (a["cigar"]["len"], int(a["alns"]["cigar_begin"][first])), (a["alns"]["score"], first)):
        arr.flags.writeable or arr.setflags(write=True)
        old = arr[idx].copy()
        arr[idx] = old + 1
        d = query_digests(res1)
        assert d[q] != d1[q] and np.array_equal(np.delete(d, q), np.delete(d1, q))
        arr[idx] = old
    assert np.array_equal(query_digests(res1), d1)

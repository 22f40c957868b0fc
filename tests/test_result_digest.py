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


def test_kernels_are_order_independent_in_the_host_model(monkeypatch):
    """The property tests/test_gpu_properties.py asserts on hardware, on the kernels' sources under the host model: split
    pipeline with the work sort and the multi-pass extension, a batch and a permutation of it, compared through the digests
    (and with the oracle's digests: the digest sees exactly what the field-by-field comparisons see)."""
    import emu_drv
    monkeypatch.setenv("MGX_EMU_SPLIT", "1")
    monkeypatch.setenv("MGX_EMU_MULTIPASS", "1")
    g, reads = make_world(4343, 19, n_reads=80)
    reads += [reads[i][:50] + reads[i + 1][40:90] for i in range(0, 20, 2)]          # chimeras: several seeds per read
    cfg = capi.config_cli(19)
    cfg.min_exact_match = 0.0
    eg = emu_drv.EmuGraph(g)

    def emu_digests(batch):
        e = emu_drv.EmuRun(eg, cfg, batch)
        assert e.error == ""
        res = capi.Results()
        emu_drv.L().emu_results(e.r, C.byref(res))
        return query_digests(res), e

    d1, keep1 = emu_digests(reads)
    perm = list(range(len(reads)))
    random.Random(2).shuffle(perm)
    d2, keep2 = emu_digests([reads[p] for p in perm])
    assert np.array_equal(d2, d1[perm])
    run, ores = _oracle_results(g, cfg, reads)
    assert np.array_equal(query_digests(ores), d1)

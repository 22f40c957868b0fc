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
    assert checked > 1000

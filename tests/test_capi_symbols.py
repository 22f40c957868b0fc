"""The C-ABI library loads on a CPU-only host and exports every symbol include/mgx.h declares.
No compute calls are made here (they need a GPU)."""
import ctypes as C
import os
import re

from metagraph_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "mgx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mgx_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    L = capi.lib()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libmgx.so does not export %s" % n
    assert L.mgx_abi_version() == capi.MGX_ABI_VERSION == 6


def test_struct_layouts_match_header():
    L = capi.lib()
    c = capi.Config()
    L.mgx_config_init_cli(C.byref(c), 31)
    ref = capi.config_cli(31)
    assert bytes(c) == bytes(ref)
    L.mgx_config_init_default(C.byref(c))
    assert bytes(c)[:96] == bytes(capi.config_default())[:96]
    assert C.sizeof(capi.Alignment) == 72 and C.sizeof(capi.CigarOp) == 8


def test_no_device_is_an_error_not_a_fallback():
    L = capi.lib()
    if L.mgx_device_count() > 0:
        return
    import numpy as np
    W = np.zeros(4, dtype=np.uint8)
    v = capi.BossView()
    v.k, v.sigma, v.n_edges = 3, 5, 3
    v.W, v.last = W.ctypes.data, W.ctypes.data
    F = (C.c_uint64 * 5)()
    v.F = C.cast(F, C.POINTER(C.c_uint64))
    h = C.c_void_p()
    rc = L.mgx_graph_create(C.byref(v), 0, C.byref(h))
    assert rc == capi.MGX_ERR_NO_DEVICE
    assert b"not available" in L.mgx_last_error()


def test_score_matrices_like_the_reference():
    """check_score_matrix_dna / check_score_matrix_dna_unit (M/tests/graph/test_aligner.cpp:28-72) on the C-ABI's
    matrix builders (DBGAlignerConfig::dna_scoring_matrix / unit_scoring_matrix, aligner_config.cpp:164-204)."""
    L = capi.lib()
    L.mgx_config_set_dna_matrix.argtypes = [C.c_void_p, C.c_int8, C.c_int8, C.c_int8]
    L.mgx_config_set_unit_matrix.argtypes = [C.c_void_p, C.c_int8]
    alphabet = "ACGTN"                                   # kAlphabetDNA5
    for build in ("dna", "unit"):
        c = capi.Config()
        L.mgx_config_init_default(C.byref(c))
        if build == "dna":
            L.mgx_config_set_dna_matrix(C.byref(c), 2, -1, -2)
        else:
            L.mgx_config_set_unit_matrix(C.byref(c), 1)
        m = lambda a, b: c.score_matrix[ord(a)][ord(b)]
        for i, a in enumerate(alphabet):
            if i + 1 != len(alphabet):
                assert m(a, a) > 0
            for b in alphabet:
                if i + 1 != len(alphabet):
                    assert m(a, a) >= m(a, b)
                assert m(a, b) == m(b, a)
        # and the Python mirrors used by the tests build the same bytes
        ref = capi.config_default()
        if build == "dna":
            capi.set_dna_matrix(ref, 2, -1, -2)
        else:
            capi.set_unit_matrix(ref, 1)
        assert bytes(c.score_matrix) == bytes(ref.score_matrix)

"""ctypes mirror of include/mgx.h and the loader for libmgx.so (the HIP product library).

PyTorch / Python are plumbing here: the product is the C-ABI library.  There is no CPU fallback;
if libmgx.so is missing the import of `lib()` raises, and every compute call fails without a GPU.
"""
import ctypes as C
import os

MGX_OK = 0
MGX_ABI_VERSION = 6
MGX_ERR_INVALID, MGX_ERR_NO_DEVICE, MGX_ERR_UNSUPPORTED, MGX_ERR_CONFIG, MGX_ERR_CAPACITY, MGX_ERR_OOM = -1, -2, -3, -4, -5, -6
OP_CHARS = "SX=DIG"


class BossView(C.Structure):
    _fields_ = [("k", C.c_uint32), ("sigma", C.c_uint32), ("n_edges", C.c_uint64),
                ("W", C.c_void_p), ("last", C.c_void_p), ("F", C.POINTER(C.c_uint64)),
                ("valid", C.c_void_p), ("mode", C.c_uint32), ("on_device", C.c_uint32)]


class BossFile(C.Structure):
    """mgx_boss_file (include/mgx.h, "files")"""
    _fields_ = [("k", C.c_uint32), ("sigma", C.c_uint32), ("mode", C.c_uint32), ("state", C.c_uint32), ("n_edges", C.c_uint64),
                ("F", C.POINTER(C.c_uint64)), ("W", C.POINTER(C.c_uint8)), ("last", C.POINTER(C.c_uint8)), ("owner", C.c_void_p)]


class Config(C.Structure):
    _fields_ = [("num_alternative_paths", C.c_uint64), ("min_seed_length", C.c_uint64),
                ("max_seed_length", C.c_uint64), ("max_num_seeds_per_locus", C.c_uint64),
                ("min_cell_score", C.c_int32), ("min_path_score", C.c_int32), ("xdrop", C.c_int32),
                ("_pad0", C.c_int32),
                ("min_exact_match", C.c_double), ("max_nodes_per_seq_char", C.c_double),
                ("max_ram_per_alignment", C.c_double), ("rel_score_cutoff", C.c_double),
                ("gap_opening_penalty", C.c_int8), ("gap_extension_penalty", C.c_int8),
                ("left_end_bonus", C.c_int8), ("right_end_bonus", C.c_int8),
                ("forward_and_reverse_complement", C.c_uint8), ("chain_alignments", C.c_uint8),
                ("post_chain_alignments", C.c_uint8), ("global_xdrop", C.c_uint8),
                ("allow_left_trim", C.c_uint8), ("no_backtrack", C.c_uint8),
                ("seed_complexity_filter", C.c_uint8), ("alignment_edit_distance", C.c_uint8),
                ("alignment_match_score", C.c_int8), ("alignment_mm_transition_score", C.c_int8),
                ("alignment_mm_transversion_score", C.c_int8), ("_pad1", C.c_uint8 * 1),
                ("score_matrix", (C.c_int8 * 128) * 128)]


class Limits(C.Structure):
    _fields_ = [("max_query_length", C.c_uint32), ("max_columns", C.c_uint32), ("max_seeds", C.c_uint32),
                ("cell_arena_bytes", C.c_uint64)]


class CigarOp(C.Structure):
    _fields_ = [("len", C.c_uint32), ("op", C.c_uint8), ("_pad", C.c_uint8 * 3)]


class Alignment(C.Structure):
    _fields_ = [("score", C.c_int32), ("offset", C.c_uint32), ("clipping", C.c_uint32),
                ("end_clipping", C.c_uint32), ("num_matches", C.c_uint32), ("n_nodes", C.c_uint32),
                ("n_cigar", C.c_uint32), ("seq_len", C.c_uint32), ("nodes_begin", C.c_uint64),
                ("cigar_begin", C.c_uint64), ("seq_begin", C.c_uint64), ("orientation", C.c_uint8),
                ("_pad", C.c_uint8 * 3), ("n_labels", C.c_uint32), ("labels_begin", C.c_uint64)]


class Results(C.Structure):
    _fields_ = [("n_queries", C.c_uint64), ("aln_begin", C.POINTER(C.c_uint64)),
                ("alignments", C.POINTER(Alignment)), ("nodes", C.POINTER(C.c_uint64)),
                ("cigar", C.POINTER(CigarOp)), ("seqs", C.POINTER(C.c_char)), ("status", C.POINTER(C.c_int32)),
                ("labels", C.POINTER(C.c_uint32))]


class Mapping(C.Structure):
    _fields_ = [("n_queries", C.c_uint64), ("node_begin", C.POINTER(C.c_uint64)),
                ("nodes_fwd", C.POINTER(C.c_uint64)), ("nodes_rc", C.POINTER(C.c_uint64))]


class Stats(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_rank_lines", C.c_uint64), ("n_select_lines", C.c_uint64),
                ("n_bit_lines", C.c_uint64), ("n_columns", C.c_uint64), ("n_extensions", C.c_uint64),
                ("n_seeds", C.c_uint64), ("n_map_lines", C.c_uint64), ("n_capacity_errors", C.c_uint64), ("phase_cycles", C.c_uint64 * 8), ("extend_cycles", C.c_uint64 * 8),
                ("seed_kernel_ms", C.c_double), ("align_kernel_ms", C.c_double),
                ("seeding_ms", C.c_double), ("sort_ms", C.c_double), ("extend_ms", C.c_double), ("n_seed_lines", C.c_uint64),
                ("n_fast_columns", C.c_uint64), ("extend_kernels", C.c_uint64), ("n_lane_reads", C.c_uint64),
                ("lane_ms", C.c_double), ("n_lane_lines", C.c_uint64), ("n_lane_columns", C.c_uint64),
                ("lane_bail_reads", C.c_uint64 * 32), ("n_capacity_retried", C.c_uint64),
                ("n_seed_lane_reads", C.c_uint64), ("seed_lane_ms", C.c_double), ("seed_lane_left_reads", C.c_uint64 * 16)]


KERNEL_GRP8, KERNEL_GRP8_PRIM, KERNEL_GRP8_ALT, KERNEL_EXT64, KERNEL_LANE, KERNEL_LAB64, KERNEL_GRP8_LAB = 1, 2, 4, 8, 16, 32, 64


def results_to_py(res):
    """Decode an mgx_results view into a list (per query) of lists of dicts."""
    out = []
    for q in range(res.n_queries):
        alns = []
        for ai in range(res.aln_begin[q], res.aln_begin[q + 1]):
            a = res.alignments[ai]
            nodes = [res.nodes[a.nodes_begin + i] for i in range(a.n_nodes)]
            cig = "".join("%d%s" % (res.cigar[a.cigar_begin + i].len, OP_CHARS[res.cigar[a.cigar_begin + i].op])
                          for i in range(a.n_cigar))
            seq = C.string_at(C.addressof(res.seqs.contents) + a.seq_begin, a.seq_len).decode() if a.seq_len else ""
            d = {"score": a.score, "offset": a.offset, "clipping": a.clipping,
                 "end_clipping": a.end_clipping, "num_matches": a.num_matches, "nodes": nodes,
                 "cigar": cig, "sequence": seq, "orientation": int(a.orientation)}
            if res.labels:                     # label-aware alignment only
                d["labels"] = [res.labels[a.labels_begin + i] for i in range(a.n_labels)]
            alns.append(d)
        out.append(alns)
    return out


def chain_alignments(config, k, res, queries):
    """mgx_chain_alignments over a Results view (host code: no GPU needed) -> decoded lists, as results_to_py"""
    import numpy as np
    blob = b"".join(q.encode() if isinstance(q, str) else bytes(q) for q in queries)
    offs = np.zeros(len(queries) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(q) for q in queries])
    store, out = C.c_void_p(), Results()
    rc = lib().mgx_chain_alignments(C.byref(config), k, C.byref(res), blob, offs.ctypes.data, C.byref(store), C.byref(out))
    if rc:
        raise RuntimeError(lib().mgx_last_error().decode())
    try:
        return results_to_py(out)
    finally:
        lib().mgx_raw_store_free(store)


def results_arrays(res):
    """numpy views of an mgx_results (valid while the owner keeps the batch alive)"""
    import numpy as np
    n = int(res.n_queries)

    def arr(ptr, count, dtype):
        if not count:
            return np.zeros(0, dtype=dtype)
        nbytes = count * np.dtype(dtype).itemsize
        return np.frombuffer((C.c_char * nbytes).from_address(C.addressof(ptr.contents)), dtype=dtype, count=count)

    aln_begin = arr(res.aln_begin, n + 1, np.uint64)
    na = int(aln_begin[-1]) if n else 0
    aln_dt = np.dtype([("score", "<i4"), ("offset", "<u4"), ("clipping", "<u4"), ("end_clipping", "<u4"),
                       ("num_matches", "<u4"), ("n_nodes", "<u4"), ("n_cigar", "<u4"), ("seq_len", "<u4"),
                       ("nodes_begin", "<u8"), ("cigar_begin", "<u8"), ("seq_begin", "<u8"), ("orientation", "u1"),
                       ("_pad", "u1", (3,)), ("n_labels", "<u4"), ("labels_begin", "<u8")])
    assert aln_dt.itemsize == C.sizeof(Alignment)
    alns = arr(res.alignments, na, aln_dt)
    tn = int(alns["n_nodes"].sum()) if na else 0
    tc = int(alns["n_cigar"].sum()) if na else 0
    ts = int(alns["seq_len"].sum()) if na else 0
    cig_dt = np.dtype([("len", "<u4"), ("op", "u1"), ("_pad", "u1", (3,))])
    return {"n": n, "aln_begin": aln_begin, "alns": alns, "nodes": arr(res.nodes, tn, np.uint64),
            "cigar": arr(res.cigar, tc, cig_dt), "seqs": arr(res.seqs, ts, np.uint8),
            "status": arr(res.status, n, np.int32)}


def count_result_mismatches(res_a, res_b):
    """Number of queries whose alignment lists differ between two mgx_results of the same batch (every field of every
    alignment: score, offset, clippings, matches, orientation, node ids, CIGAR runs, path spelling).  Whole-array
    comparison first; the per-query walk only runs when something differs."""
    import numpy as np
    a, b = results_arrays(res_a), results_arrays(res_b)
    assert a["n"] == b["n"]
    fields = ["score", "offset", "clipping", "end_clipping", "num_matches", "n_nodes", "n_cigar", "seq_len", "orientation"]
    same = np.array_equal(a["aln_begin"], b["aln_begin"]) and len(a["alns"]) == len(b["alns"]) \
        and all(np.array_equal(a["alns"][f], b["alns"][f]) for f in fields) \
        and np.array_equal(a["nodes"], b["nodes"]) and np.array_equal(a["cigar"]["len"], b["cigar"]["len"]) \
        and np.array_equal(a["cigar"]["op"], b["cigar"]["op"]) and np.array_equal(a["seqs"], b["seqs"])
    if same:
        return 0
    pa, pb = results_to_py(res_a), results_to_py(res_b)
    return sum(1 for x, y in zip(pa, pb) if x != y)


_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("MGX_LIB_PATH") or os.path.join(_ROOT, "metagraph_amd", "_build", "libmgx.so")   # override: A/B builds
_lib = None


def lib():
    """Load libmgx.so (raises OSError if it has not been built — there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("libmgx.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(needs hipcc); the aligner has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    L.mgx_last_error.restype = C.c_char_p
    L.mgx_abi_version.restype = C.c_uint32
    if L.mgx_abi_version() != MGX_ABI_VERSION:          # struct layouts (mgx_stats, mgx_config) are part of the ABI
        raise OSError("libmgx.so has ABI version %d, this binding was written for %d: rebuild" % (L.mgx_abi_version(), MGX_ABI_VERSION))
    L.mgx_annotation_create.argtypes = [C.c_uint64, C.c_uint32, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]
    L.mgx_annotation_destroy.argtypes = [C.c_void_p]
    L.mgx_annotation_create_sparse.argtypes = [C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.mgx_labeled_aligner_create.argtypes = [C.c_void_p, C.POINTER(Config), C.POINTER(Limits), C.c_void_p, C.POINTER(C.c_void_p)]
    L.mgx_format_tsv_labeled.argtypes = [C.POINTER(Results), C.c_uint64, C.c_char_p, C.c_char_p, C.c_size_t, C.c_int32,
                                         C.POINTER(C.c_char_p), C.c_uint32, C.c_char_p, C.c_size_t]
    L.mgx_format_tsv_labeled.restype = C.c_size_t
    L.mgx_annotation_device_bytes.argtypes = [C.c_void_p]
    L.mgx_annotation_device_bytes.restype = C.c_uint64
    L.mgx_annotation_get_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int,
                                          C.POINTER(C.c_uint64)]
    L.mgx_device_count.restype = C.c_int
    L.mgx_boss_file_read.argtypes = [C.c_char_p, C.POINTER(BossFile)]
    L.mgx_boss_file_free.argtypes = [C.POINTER(BossFile)]
    L.mgx_edgemask_read.argtypes = [C.c_char_p, C.c_uint32, C.c_uint64, C.c_void_p]
    L.mgx_graph_load_dbg.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
    L.mgx_column_file_read.argtypes = [C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_void_p)]
    L.mgx_column_file_free.argtypes = [C.c_void_p]
    L.mgx_column_file_num_rows.argtypes = [C.c_void_p]
    L.mgx_column_file_num_rows.restype = C.c_uint64
    L.mgx_column_file_num_labels.argtypes = [C.c_void_p]
    L.mgx_column_file_num_labels.restype = C.c_uint32
    L.mgx_column_file_label.argtypes = [C.c_void_p, C.c_uint32]
    L.mgx_column_file_label.restype = C.c_char_p
    L.mgx_column_file_col_begin.argtypes = [C.c_void_p]
    L.mgx_column_file_col_begin.restype = C.POINTER(C.c_uint64)
    L.mgx_column_file_rows.argtypes = [C.c_void_p]
    L.mgx_column_file_rows.restype = C.POINTER(C.c_uint64)
    L.mgx_annotation_create_from_file.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.mgx_graph_create.argtypes = [C.POINTER(BossView), C.c_int, C.POINTER(C.c_void_p)]
    L.mgx_graph_destroy.argtypes = [C.c_void_p]
    L.mgx_graph_k.argtypes = [C.c_void_p]
    L.mgx_graph_k.restype = C.c_uint32
    L.mgx_graph_max_index.argtypes = [C.c_void_p]
    L.mgx_graph_max_index.restype = C.c_uint64
    L.mgx_graph_device_bytes.argtypes = [C.c_void_p]
    L.mgx_graph_num_edges.argtypes = [C.c_void_p]
    L.mgx_graph_num_edges.restype = C.c_uint64
    L.mgx_graph_mode.argtypes = [C.c_void_p]
    L.mgx_graph_mode.restype = C.c_uint32
    L.mgx_graph_device_bytes.restype = C.c_uint64
    L.mgx_aligner_create.argtypes = [C.c_void_p, C.POINTER(Config), C.POINTER(Limits), C.POINTER(C.c_void_p)]
    L.mgx_aligner_destroy.argtypes = [C.c_void_p]
    L.mgx_aligner_get_config.argtypes = [C.c_void_p, C.POINTER(Config)]
    L.mgx_aligner_get_limits.argtypes = [C.c_void_p, C.POINTER(Limits)]
    L.mgx_align_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(Results)]
    L.mgx_align_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
    L.mgx_fetch_results.argtypes = [C.c_void_p, C.POINTER(Results)]
    L.mgx_aligner_keep_seeds.argtypes = [C.c_void_p, C.c_int]
    L.mgx_device_results.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.mgx_device_stream_capacity.argtypes = [C.c_void_p]
    L.mgx_device_stream_capacity.restype = C.c_uint64
    L.mgx_results_from_raw.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(Results)]
    L.mgx_chain_alignments.argtypes = [C.POINTER(Config), C.c_uint32, C.POINTER(Results), C.c_char_p, C.c_void_p,
                                       C.POINTER(C.c_void_p), C.POINTER(Results)]
    L.mgx_chain_alignments.restype = C.c_int
    L.mgx_results_from_raw_labeled.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_void_p), C.POINTER(Results)]
    L.mgx_raw_store_free.argtypes = [C.c_void_p]
    # the RCCL gather of the device results (include/mgx.h, csrc/mgx_gather.hip)
    L.mgx_gather_unique_id_bytes.restype = C.c_uint64
    L.mgx_gather_unique_id.argtypes = [C.c_void_p, C.c_uint64]
    L.mgx_gather_create_rank.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.mgx_gather_create_comm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.mgx_gather_create_local.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.mgx_gather_destroy.argtypes = [C.c_void_p]
    L.mgx_gather_destroy.restype = None
    L.mgx_gather_world.argtypes = [C.c_void_p]
    L.mgx_gather_rank.argtypes = [C.c_void_p]
    L.mgx_gather_start.argtypes = [C.c_void_p, C.c_void_p]
    L.mgx_gather_finish.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.mgx_map_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(Mapping)]
    L.mgx_aligner_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.mgx_config_init_default.argtypes = [C.POINTER(Config)]
    L.mgx_config_init_cli.argtypes = [C.POINTER(Config), C.c_uint32]
    L.mgx_config_set_dna_matrix.argtypes = [C.POINTER(Config), C.c_int8, C.c_int8, C.c_int8]
    L.mgx_config_set_unit_matrix.argtypes = [C.POINTER(Config), C.c_int8]
    L.mgx_config_set_scoring_matrix.argtypes = [C.POINTER(Config)]
    L.mgx_limits_init_default.argtypes = [C.POINTER(Limits), C.c_uint32]
    L.mgx_format_tsv.argtypes = [C.POINTER(Results), C.c_uint64, C.c_char_p, C.c_char_p, C.c_size_t,
                                 C.c_int32, C.c_char_p, C.c_size_t]
    L.mgx_format_tsv.restype = C.c_size_t
    L.mgx_format_json.argtypes = [C.POINTER(Results), C.c_uint64, C.c_char_p, C.c_char_p, C.c_size_t,
                                  C.c_uint32, C.c_char_p, C.c_size_t]
    L.mgx_format_json.restype = C.c_size_t
    _lib = L
    return L


# ---- host-side config helpers that do not need the library (same tables as mgx_config_init_*) ----
INT32_MAX = 2**31 - 1
NINF = -2**31 + 100
UINT64_MAX = 2**64 - 1
DBL_MAX = 1.7976931348623157e308


def config_default():
    """DBGAlignerConfig{} (graph/alignment/aligner_config.hpp:23-54)."""
    c = Config()
    c.num_alternative_paths = 1
    c.max_num_seeds_per_locus = UINT64_MAX
    c.min_cell_score = NINF
    c.min_path_score = 0
    c.xdrop = INT32_MAX
    c.min_exact_match = 0.0
    c.max_nodes_per_seq_char = DBL_MAX
    c.max_ram_per_alignment = DBL_MAX
    c.rel_score_cutoff = 0.0
    c.gap_opening_penalty = -5
    c.gap_extension_penalty = -2
    c.forward_and_reverse_complement = 1
    c.global_xdrop = 1
    c.allow_left_trim = 1
    c.seed_complexity_filter = 1
    return c


def set_dna_matrix(c, match, transition, transversion):
    """DBGAlignerConfig::dna_scoring_matrix (aligner_config.cpp:164-183)."""
    for i in range(128):
        for j in range(128):
            c.score_matrix[i][j] = transversion
    for a, b in (("A", "G"), ("G", "A"), ("C", "T"), ("T", "C")):
        c.score_matrix[ord(a)][ord(b)] = transition
    for a in "ACGT":
        c.score_matrix[ord(a)][ord(a)] = match


def set_unit_matrix(c, match):
    """DBGAlignerConfig::unit_scoring_matrix over "ACGT" (aligner_config.cpp:185-204)."""
    for i in range(128):
        for j in range(128):
            c.score_matrix[i][j] = -match
    for a in "ACGT":
        c.score_matrix[ord(a)][ord(a)] = match


def config_cli(k):
    """`metagraph align` defaults (cli/config/config.hpp:114-145 via cli/align.cpp:33-69)."""
    c = config_default()
    c.min_seed_length = min(19, k)
    c.max_seed_length = UINT64_MAX
    c.max_num_seeds_per_locus = 1000
    c.min_path_score = 0
    c.xdrop = 27
    c.min_exact_match = 0.7
    c.max_nodes_per_seq_char = 5.0
    c.max_ram_per_alignment = 200.0
    c.rel_score_cutoff = 0.95
    c.gap_opening_penalty = -6
    c.gap_extension_penalty = -2
    c.left_end_bonus = 5
    c.right_end_bonus = 5
    c.alignment_edit_distance = 0
    c.alignment_match_score, c.alignment_mm_transition_score, c.alignment_mm_transversion_score = 2, 3, 3
    set_dna_matrix(c, 2, -3, -3)
    return c


def format_json(res, qi, header, query, k):
    """`metagraph align --json` lines of query qi of an mgx_results view (host-side formatting, no GPU involved)."""
    q = query if isinstance(query, bytes) else query.encode("latin-1")
    h = header if isinstance(header, bytes) else header.encode()
    n = lib().mgx_format_json(C.byref(res), qi, h, q, len(q), k, None, 0)
    buf = C.create_string_buffer(n + 1)
    lib().mgx_format_json(C.byref(res), qi, h, q, len(q), k, buf, n + 1)
    return buf.value.decode("latin-1")

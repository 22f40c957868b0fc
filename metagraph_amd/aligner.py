"""Thin Python plumbing over libmgx.so (ctypes).  Mirrors the reference's operator surface for the
alignment path: `Graph` ~ DBGSuccinct (BOSS view upload), `Aligner.align_batch` ~
IDBGAligner::align_batch (graph/alignment/dbg_aligner.hpp:20-39).  No compute happens in Python and
there is no CPU fallback: every call below fails loudly without the HIP library / a GPU.
"""
import ctypes as C

import numpy as np

from . import capi


class MgxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mgx error %d: %s" % (code, msg))
        self.code = code


def _check(rc):
    if rc != capi.MGX_OK:
        raise MgxError(rc, capi.lib().mgx_last_error().decode())


def pack_queries(queries):
    offs = np.zeros(len(queries) + 1, dtype=np.uint64)
    bs = [q if isinstance(q, bytes) else q.encode("latin-1") for q in queries]
    for i, b in enumerate(bs):
        offs[i + 1] = offs[i] + len(b)
    return b"".join(bs), offs


def read_boss_file(path, arrays=True):
    """mgx_boss_file_read (host only, no GPU): k, sigma, mode, state, n_edges, F and — with arrays — W / last of a `.dbg` file."""
    L = capi.lib()
    f = capi.BossFile()
    _check(L.mgx_boss_file_read(str(path).encode(), C.byref(f)))
    try:
        out = {"k": f.k, "sigma": f.sigma, "mode": f.mode, "state": f.state, "n_edges": f.n_edges, "F": [int(f.F[i]) for i in range(f.sigma)]}
        if arrays:
            out["W"] = np.ctypeslib.as_array(f.W, shape=(f.n_edges + 1,)).copy()
            out["last"] = np.ctypeslib.as_array(f.last, shape=(f.n_edges + 1,)).copy()
        return out
    finally:
        L.mgx_boss_file_free(C.byref(f))


def read_edgemask(path, state, n_edges):
    """mgx_edgemask_read (host only): the valid-edge bytes (mgx_boss_view.valid) of the `.edgemask` next to a `.dbg`"""
    out = np.zeros(n_edges + 1, dtype=np.uint8)
    _check(capi.lib().mgx_edgemask_read(str(path).encode(), state, n_edges, out.ctypes.data))
    return out


def read_column_files(paths):
    """mgx_column_file_read (host only): (n_rows, label names, col_begin, rows) of one or several `.column.annodbg` files."""
    L = capi.lib()
    arr = (C.c_char_p * len(paths))(*[str(p).encode() for p in paths])
    h = C.c_void_p()
    _check(L.mgx_column_file_read(arr, len(paths), C.byref(h)))
    try:
        n = L.mgx_column_file_num_labels(h)
        names = [L.mgx_column_file_label(h, j).decode() for j in range(n)]
        cb = np.ctypeslib.as_array(L.mgx_column_file_col_begin(h), shape=(n + 1,)).copy()
        rows = np.ctypeslib.as_array(L.mgx_column_file_rows(h), shape=(int(cb[-1]),)).copy() if int(cb[-1]) else np.zeros(0, dtype=np.uint64)
        return int(L.mgx_column_file_num_rows(h)), names, cb, rows
    finally:
        L.mgx_column_file_free(h)


class Graph:
    """A BOSS table on the GPU.  W/last: uint8 arrays of n_edges + 1 entries (slot 0 unused)."""

    def __init__(self, k, W, last, F, valid=None, device=0, on_device=False, mode=0):
        L = capi.lib()
        v = capi.BossView()
        v.k = k
        v.sigma = 5
        if on_device:      # (ptr, n_entries) pairs for W / last / valid
            v.n_edges = W[1] - 1
            v.W, v.last = W[0], last[0]
            v.valid = valid[0] if valid is not None else None
        else:
            W = np.ascontiguousarray(W, dtype=np.uint8)
            last = np.ascontiguousarray(last, dtype=np.uint8)
            v.n_edges = len(W) - 1
            v.W, v.last = W.ctypes.data, last.ctypes.data
            if valid is not None:
                valid = np.ascontiguousarray(valid, dtype=np.uint8)
                v.valid = valid.ctypes.data
        Fc = (C.c_uint64 * 5)(*[int(x) for x in F])
        v.F = C.cast(Fc, C.POINTER(C.c_uint64))
        v.mode = mode
        v.on_device = 1 if on_device else 0
        self.h = C.c_void_p()
        _check(L.mgx_graph_create(C.byref(v), device, C.byref(self.h)))
        self.k = k
        self.n_edges = v.n_edges

    @classmethod
    def load(cls, path, device=0):
        """mgx_graph_load_dbg: a `.dbg` file written by the reference (DBGSuccinct::load, dbg_succinct.cpp:690-785)."""
        self = cls.__new__(cls)
        self.h = C.c_void_p()
        _check(capi.lib().mgx_graph_load_dbg(str(path).encode(), device, C.byref(self.h)))
        self.k = capi.lib().mgx_graph_k(self.h)
        self.n_edges, self.mode = capi.lib().mgx_graph_num_edges(self.h), capi.lib().mgx_graph_mode(self.h)
        return self

    def close(self):
        if getattr(self, "h", None) and capi is not None:      # capi is None during interpreter shutdown
            capi.lib().mgx_graph_destroy(self.h)
            self.h = None

    __del__ = close

    @property
    def device_bytes(self):
        return capi.lib().mgx_graph_device_bytes(self.h)


class Annotation:
    """The label matrix of an annotated graph on the GPU (mgx_annotation_create): columns[j] = the bit vector of label j over the
    rows (row = node - 1), 64 rows per uint64 word."""

    def __init__(self, n_rows, columns, device=0):
        self._cols = [np.ascontiguousarray(c, dtype=np.uint64) for c in columns]
        ptrs = (C.c_void_p * max(1, len(self._cols)))(*[c.ctypes.data for c in self._cols])
        self.h = C.c_void_p()
        self.n_rows, self.n_labels = n_rows, len(self._cols)
        _check(capi.lib().mgx_annotation_create(n_rows, len(self._cols), ptrs, device, C.byref(self.h)))

    @classmethod
    def from_sparse(cls, n_rows, col_begin, rows, device=0, on_device=False):
        """mgx_annotation_create_sparse: rows[col_begin[j] : col_begin[j + 1]] = the rows with label j (a ColumnCompressed
        annotation's content).  col_begin: host uint64 array of n_labels + 1 entries; rows: host uint64 array, or a device
        pointer with on_device=True."""
        self = cls.__new__(cls)
        cb = np.ascontiguousarray(col_begin, dtype=np.uint64)
        self._cols = [cb]
        rp = rows
        if not on_device:
            r = np.ascontiguousarray(rows, dtype=np.uint64)
            self._cols.append(r)
            rp = r.ctypes.data
        self.h = C.c_void_p()
        self.n_rows, self.n_labels = n_rows, len(cb) - 1
        _check(capi.lib().mgx_annotation_create_sparse(n_rows, len(cb) - 1, cb.ctypes.data, rp, 1 if on_device else 0, device, C.byref(self.h)))
        return self

    @classmethod
    def load(cls, paths, device=0):
        """`.column.annodbg` files written by the reference (ColumnCompressed::load / merge_load); label names in .labels."""
        if isinstance(paths, (str, bytes)) or hasattr(paths, "__fspath__"):
            paths = [paths]
        n_rows, names, cb, rows = read_column_files(paths)
        self = cls.from_sparse(n_rows, cb, rows, device=device)
        self.labels = names
        return self

    @property
    def device_bytes(self):
        return capi.lib().mgx_annotation_device_bytes(self.h)

    def close(self):
        if getattr(self, "h", None) and capi is not None:
            capi.lib().mgx_annotation_destroy(self.h)
            self.h = None

    __del__ = close


class Aligner:
    """DBGAligner<> on the GPU (default seeder/extender); with `annotation`: LabeledAligner<> (aligner_labeled.hpp:125-127)."""

    # kernel-selection options (mgx_aligner_set_pipeline "key=value") every new aligner starts with: all of them give the
    # same alignments; the parity suite sets this to run a kernel the automatic choice would not pick for its batch sizes
    default_options = ()

    def __init__(self, graph, config, limits=None, annotation=None):
        self.graph = graph
        self.annotation = annotation
        self.h = C.c_void_p()
        if annotation is not None:
            _check(capi.lib().mgx_labeled_aligner_create(graph.h, C.byref(config), C.byref(limits) if limits is not None else None,
                                                         annotation.h, C.byref(self.h)))
        else:
            _check(capi.lib().mgx_aligner_create(graph.h, C.byref(config), C.byref(limits) if limits is not None else None,
                                                 C.byref(self.h)))
        for opt in Aligner.default_options:
            self.set_pipeline(opt)

    def close(self):
        if getattr(self, "h", None) and capi is not None:
            capi.lib().mgx_aligner_destroy(self.h)
            self.h = None

    __del__ = close

    def get_config(self):
        c = capi.Config()
        _check(capi.lib().mgx_aligner_get_config(self.h, C.byref(c)))
        return c

    def align_batch(self, queries):
        """-> (list per query of alignment dicts, list of status codes)"""
        blob, offs = pack_queries(queries)
        res = capi.Results()
        _check(capi.lib().mgx_align_batch(self.h, blob, offs.ctypes.data, len(queries), 0, C.byref(res)))
        return capi.results_to_py(res), [res.status[i] for i in range(len(queries))]

    def align_device(self, seqs_ptr, offsets_ptr, n):
        """Reads already in HBM (device pointers); results stay on the device until fetch()."""
        _check(capi.lib().mgx_align_batch_device(self.h, seqs_ptr, offsets_ptr, n, 1))

    def fetch(self):
        res = capi.Results()
        _check(capi.lib().mgx_fetch_results(self.h, C.byref(res)))
        return res

    def map_batch(self, queries):
        blob, offs = pack_queries(queries)
        m = capi.Mapping()
        _check(capi.lib().mgx_map_batch(self.h, blob, offs.ctypes.data, len(queries), 0, C.byref(m)))
        out = []
        for q in range(len(queries)):
            b, e = m.node_begin[q], m.node_begin[q + 1]
            out.append(([m.nodes_fwd[i] for i in range(b, e)], [m.nodes_rc[i] for i in range(b, e)]))
        return out

    def set_pipeline(self, name):
        """Kernel selection: 'split8' (the pipeline), 'general' / 'chain' (extension chain path off / on), 'key=value'
        options (include/mgx.h, mgx_aligner_set_pipeline); results never depend on it."""
        L = capi.lib()
        L.mgx_aligner_set_pipeline.argtypes = [C.c_void_p, C.c_char_p]
        _check(L.mgx_aligner_set_pipeline(self.h, name.encode()))

    def keep_seeds(self, keep=True):
        capi.lib().mgx_aligner_keep_seeds(self.h, int(keep))

    def seed_info(self, n):
        from . import _seedinfo
        return _seedinfo.fetch(self, n)

    def stats(self):
        s = capi.Stats()
        _check(capi.lib().mgx_aligner_stats(self.h, C.byref(s)))
        d = {f[0]: getattr(s, f[0]) for f in capi.Stats._fields_}
        d["phase_cycles"] = list(s.phase_cycles)
        d["extend_cycles"] = list(s.extend_cycles)
        d["lane_bail_reads"] = {i: int(v) for i, v in enumerate(s.lane_bail_reads) if v}
        d["seed_lane_left_reads"] = {i: int(v) for i, v in enumerate(s.seed_lane_left_reads) if v}
        return d

    def format_tsv(self, res, qi, header, query):
        q = query if isinstance(query, bytes) else query.encode("latin-1")
        cfg = self.get_config()
        n = capi.lib().mgx_format_tsv(C.byref(res), qi, header.encode(), q, len(q), cfg.min_path_score, None, 0)
        buf = C.create_string_buffer(n + 1)
        capi.lib().mgx_format_tsv(C.byref(res), qi, header.encode(), q, len(q), cfg.min_path_score, buf, n + 1)
        return buf.value.decode("latin-1")

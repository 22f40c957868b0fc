"""ctypes driver for tests/emu/_build/libmgxemu.so — the host model of the kernels' wave programs.
TEST INFRASTRUCTURE ONLY (CPU-only CI); the product path is libmgx.so on a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np

from metagraph_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_L = None


def L():
    global _L
    if _L is None:
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")], check=True)
        # MGX_EMU_WAVE=16 / 8 selects the model of the sub-wave-group kernels (default: one 64-lane wavefront)
        suffix = {"16": "_w16", "8": "_w8"}.get(os.environ.get("MGX_EMU_WAVE", ""), "")
        if os.environ.get("MGX_EMU_TRACE_LIB"):                 # tools/traffic_model.py: the traced build (make trace)
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "trace"], check=True, stderr=subprocess.DEVNULL)
            suffix = "_trace_w8"
        if os.environ.get("MGX_EMU_LANE_CHECK"):                # tests/test_lane_column.py: chain_step cross-checks lane_column()
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "lanecheck"], check=True)
            suffix = "_lanecheck_w8"
        _L = C.CDLL(os.environ.get("MGX_EMU_LIB") or os.path.join(ROOT, "tests", "emu", "_build", "libmgxemu%s.so" % suffix))   # (MGX_EMU_LIB: an instrumented build)
        _L.emu_graph_create.restype = C.c_void_p
        _L.emu_graph_create.argtypes = [C.POINTER(capi.BossView)]
        _L.emu_graph_free.argtypes = [C.c_void_p]
        _L.emu_fwd.restype = C.c_uint64
        _L.emu_fwd.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        _L.emu_bwd.restype = C.c_uint64
        _L.emu_bwd.argtypes = [C.c_void_p, C.c_uint64]
        _L.emu_first_char.restype = C.c_uint32
        _L.emu_first_char.argtypes = [C.c_void_p, C.c_uint64]
        _L.emu_terminus.argtypes = [C.c_void_p, C.c_uint64]
        _L.emu_outgoing.restype = C.c_uint32
        _L.emu_outgoing.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64), C.c_char_p]
        _L.emu_canon_children.restype = C.c_uint32
        _L.emu_canon_children.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_char_p, C.POINTER(C.c_int)]
        _L.emu_terminus_primary.argtypes = [C.c_void_p, C.c_uint64]
        _L.emu_is_low_complexity.argtypes = [C.c_char_p, C.c_uint32]
        _L.emu_maybe_low_complexity.argtypes = [C.c_char_p, C.c_uint32]
        _L.emu_align.restype = C.c_void_p
        _L.emu_align.argtypes = [C.c_void_p, C.POINTER(capi.Config), C.POINTER(capi.Limits), C.c_char_p,
                                 C.POINTER(C.c_uint64), C.c_uint64, C.c_int]
        _L.emu_annotation_create.restype = C.c_void_p
        _L.emu_annotation_create.argtypes = [C.c_uint64, C.c_uint32, C.POINTER(C.c_void_p)]
        _L.emu_annotation_free.argtypes = [C.c_void_p]
        _L.emu_align_anno.restype = C.c_void_p
        _L.emu_align_anno.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(capi.Config), C.POINTER(capi.Limits), C.c_char_p,
                                      C.POINTER(C.c_uint64), C.c_uint64, C.c_int]
        _L.emu_error.restype = C.c_char_p
        _L.emu_error.argtypes = [C.c_void_p]
        _L.emu_results.argtypes = [C.c_void_p, C.POINTER(capi.Results)]
        _L.emu_mapping.argtypes = [C.c_void_p, C.POINTER(capi.Mapping)]
        _L.emu_seed_info.restype = C.c_uint32
        _L.emu_seed_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        _L.emu_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        _L.emu_free.argtypes = [C.c_void_p]
    return _L


def boss_view(k, W, last, F, valid=None, on_device=0, mode=0):
    """numpy arrays -> (BossView, keepalive)"""
    v = capi.BossView()
    v.k = k
    v.sigma = 5
    v.n_edges = len(W) - 1
    v.W = W.ctypes.data
    v.last = last.ctypes.data
    Fc = (C.c_uint64 * 5)(*[int(x) for x in F])
    v.F = C.cast(Fc, C.POINTER(C.c_uint64))
    v.valid = valid.ctypes.data if valid is not None else None
    v.mode = mode
    v.on_device = on_device
    return v, (W, last, Fc, valid)


class EmuGraph:
    def __init__(self, orc_graph, mode=0):
        W, last, F, valid = orc_graph.export()
        self.view, self._keep = boss_view(orc_graph.k, W, last, F, valid, mode=mode)
        self.h = L().emu_graph_create(C.byref(self.view))
        self.k = orc_graph.k

    def __del__(self):
        if getattr(self, "h", None):
            L().emu_graph_free(self.h)
            self.h = None

    def outgoing(self, v, rc=False):
        nodes = (C.c_uint64 * 8)()
        chars = C.create_string_buffer(8)
        n = L().emu_outgoing(self.h, v, int(rc), nodes, chars)
        return [(nodes[i], chars.raw[i:i + 1].decode()) for i in range(n)]


    def canon_children(self, v):
        """CanonicalDBG::call_outgoing_kmers of wrapper id v (PRIMARY graphs) -> ([(node, char)], sentinel flag)"""
        nodes = (C.c_uint64 * 8)()
        chars = C.create_string_buffer(8)
        sent = C.c_int(0)
        n = L().emu_canon_children(self.h, v, nodes, chars, C.byref(sent))
        return [(nodes[i], chars.raw[i:i + 1].decode()) for i in range(n)], bool(sent.value)

    def terminus_primary(self, v):
        return bool(L().emu_terminus_primary(self.h, v))


class EmuAnnotation:
    """The label matrix in the device's row-major form, from an orc.Annotation (its column bit vectors)."""

    def __init__(self, orc_annotation):
        self.n_labels = orc_annotation.n_labels
        self._cols = [np.ascontiguousarray(orc_annotation.column_words(j), dtype=np.uint64) for j in range(self.n_labels)]
        ptrs = (C.c_void_p * max(1, self.n_labels))(*[c.ctypes.data for c in self._cols])
        self.n_rows = orc_annotation.graph.n_edges
        self.h = L().emu_annotation_create(self.n_rows, self.n_labels, ptrs)

    def __del__(self):
        if getattr(self, "h", None):
            L().emu_annotation_free(self.h)
            self.h = None


class EmuRun:
    def __init__(self, graph, config, queries, limits=None, map_only=False, annotation=None):
        from orc import pack_queries
        blob, offs = pack_queries(queries)
        self._keep = (blob, offs, annotation)
        self.n = len(queries)
        self.r = L().emu_align_anno(graph.h, annotation.h if annotation is not None else None, C.byref(config),
                                    C.byref(limits) if limits is not None else None, blob,
                                    offs.ctypes.data_as(C.POINTER(C.c_uint64)), self.n, int(map_only))
        self.error = L().emu_error(self.r).decode()

    def __del__(self):
        if getattr(self, "r", None):
            L().emu_free(self.r)
            self.r = None

    def retried(self):
        """reads that pass 1 of the two-pass extension handed to pass 2 (split pipeline only)"""
        L().emu_retried.restype = C.c_uint64
        L().emu_retried.argtypes = [C.c_void_p]
        return L().emu_retried(self.r)

    def raw(self):
        """-> (headers bytes, stream bytes): the device-layout records mgx_device_results exposes on the GPU"""
        hp, n, sp, w = C.c_void_p(), C.c_uint64(), C.c_void_p(), C.c_uint64()
        L().emu_raw.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L().emu_raw(self.r, C.byref(hp), C.byref(n), C.byref(sp), C.byref(w))
        return C.string_at(hp.value, 64 * n.value) if n.value else b"", C.string_at(sp.value, 4 * w.value) if w.value else b""

    def results(self):
        v = capi.Results()
        L().emu_results(self.r, C.byref(v))
        return capi.results_to_py(v), [v.status[i] for i in range(self.n)]

    def mapping(self):
        m = capi.Mapping()
        L().emu_mapping(self.r, C.byref(m))
        out = []
        for q in range(self.n):
            b, e = m.node_begin[q], m.node_begin[q + 1]
            out.append(([m.nodes_fwd[i] for i in range(b, e)], [m.nodes_rc[i] for i in range(b, e)]))
        return out

    def map_side(self):
        """k_map's side outputs per strand: (match-length bytes, (rl, ru) words)"""
        out = []
        L().emu_map_side.restype = C.c_uint64
        L().emu_map_side.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        for s in (0, 1):
            lp, rp = C.c_void_p(), C.c_void_p()
            n = L().emu_map_side(self.r, s, C.byref(lp), C.byref(rp))
            out.append((C.string_at(lp.value, n) if n else b"", C.string_at(rp.value, 8 * n) if n else b""))
        return out

    def seed_info(self):
        info = (C.c_uint32 * (6 * self.n))()
        ms = L().emu_seed_info(self.r, info, None)
        seeds = (C.c_uint32 * (self.n * 2 * ms * 4))()
        L().emu_seed_info(self.r, info, seeds)
        return decode_seed_info(self.n, ms, info, seeds)

    def lane_stats(self):
        """MGX_EMU_LANE=1 runs: (whether the lane-per-read path ran for the batch, reads it finished)"""
        a = (C.c_uint64 * 2)()
        L().emu_lane_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L().emu_lane_stats(self.r, a)
        return bool(a[0]), int(a[1])

    def lane_reasons(self):
        """per read: 0 = finished by the lane-per-read path, else the LANE_BAIL code that sent it on"""
        a = (C.c_uint8 * max(1, self.n))()
        L().emu_lane_reasons.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_uint64]
        L().emu_lane_reasons(self.r, a, self.n)
        return list(a)[:self.n]

    def seedlane_stats(self):
        """MGX_EMU_SEEDLANE=1 runs: (whether the lane-per-read seeder ran, reads it finished, {reason: reads it left})"""
        a = (C.c_uint64 * 18)()
        L().emu_seedlane_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L().emu_seedlane_stats(self.r, a)
        return bool(a[0]), int(a[1]), {i: int(a[2 + i]) for i in range(16) if a[2 + i]}

    def lane_bails(self):
        """reads the lane-per-read path sent on to the wave program, by LANE_BAIL code (lane_read.hpp)"""
        a = (C.c_uint64 * 32)()
        L().emu_lane_bails.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L().emu_lane_bails(self.r, a)
        return {i: int(a[i]) for i in range(32) if a[i]}

    def stats(self):
        a = (C.c_uint64 * 8)()
        L().emu_stats(self.r, a)
        return dict(zip(["rank_lines", "select_lines", "bit_lines", "columns", "extensions", "seeds", "capacity_errors",
                         "fast_columns"], list(a)))


def decode_seed_info(n, ms, info, seeds):
    """-> per read: {'num_matches': (f, r), 'seeds': (list_f, list_r), 'n_extensions', 'n_columns'};
    a seed = (clipping, length, offset, n_nodes or the sub-k node)"""
    out = []
    arr = np.ctypeslib.as_array(seeds).reshape(n, 2, ms, 4) if n else None
    for i in range(n):
        nf, nr = info[6 * i + 2], info[6 * i + 3]
        sl = []
        for s, cnt in ((0, nf), (1, nr)):
            sl.append([tuple(int(x) for x in arr[i, s, j]) for j in range(cnt)])
        out.append({"num_matches": (info[6 * i], info[6 * i + 1]), "seeds": tuple(sl),
                    "n_extensions": info[6 * i + 4], "n_columns": info[6 * i + 5]})
    return out


def oracle_seeds_as_tuples(seed_list):
    """orc.AlignRun.seeds(strand)[q][0] -> same tuple form as decode_seed_info"""
    return [(s["clipping"], s["length"], s["offset"], s["nodes"][0] if s["offset"] else len(s["nodes"])) for s in seed_list]

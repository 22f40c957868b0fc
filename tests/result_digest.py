"""One 64-bit digest per query of an mgx_results (every field of every alignment, order-sensitive inside a query), vectorised
with numpy so that whole batches of millions of reads can be compared without an oracle: used by the size-independent property
tests (the results of a read do not depend on where it stands in the batch, nor on what ran before).  Test infrastructure."""
import numpy as np

from metagraph_amd import capi

_M1, _M2, _M3, _M4 = (np.uint64(0x9E3779B97F4A7C15), np.uint64(0xC2B2AE3D27D4EB4F), np.uint64(0x165667B19E3779F9),
                      np.uint64(0x27D4EB2F165667C5))


def _segment_weighted_sums(values, counts):
    """sum over each segment of values[t] * (2 t + 1), t = position inside the segment; uint64 arithmetic modulo 2^64"""
    counts = counts.astype(np.int64)
    total = int(counts.sum())
    out = np.zeros(len(counts), dtype=np.uint64)
    if not total:
        return out
    begin = np.concatenate(([0], np.cumsum(counts)[:-1]))
    t = np.arange(total, dtype=np.int64) - np.repeat(begin, counts)
    w = values.astype(np.uint64) * (2 * t + 1).astype(np.uint64)
    c = np.concatenate(([np.uint64(0)], np.cumsum(w, dtype=np.uint64)))
    end = begin + counts
    return c[end] - c[begin]


def query_digests(res):
    """-> uint64 array, one digest per query (0 alignments -> a digest of the status only)"""
    a = capi.results_arrays(res)
    alns = a["alns"]
    with np.errstate(over="ignore"):
        h = (alns["score"].astype(np.int64).astype(np.uint64) * _M1 + alns["offset"].astype(np.uint64) * _M2
             + alns["clipping"].astype(np.uint64) * _M3 + alns["end_clipping"].astype(np.uint64) * _M4
             + alns["num_matches"].astype(np.uint64) * np.uint64(0x9FB21C651E98DF25) + alns["orientation"].astype(np.uint64) * np.uint64(0xD6E8FEB86659FD93))
        h = h + _segment_weighted_sums(a["nodes"], alns["n_nodes"]) * _M2
        cig = a["cigar"]["len"].astype(np.uint64) * np.uint64(8) + a["cigar"]["op"].astype(np.uint64)
        h = h + _segment_weighted_sums(cig, alns["n_cigar"]) * _M3
        h = h + _segment_weighted_sums(a["seqs"], alns["seq_len"]) * _M4
        h ^= h >> np.uint64(29)
        h *= _M1
        per_query = np.diff(a["aln_begin"].astype(np.int64))
        d = _segment_weighted_sums(h, per_query) + a["status"].astype(np.int64).astype(np.uint64) * _M4 + per_query.astype(np.uint64)
    return d

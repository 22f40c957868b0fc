"""Gather of complete alignments to rank 0 (the reference's "gather" is the whole output line of every read,
cli/align.cpp:469-473; here: one process per GPU, results in the device layout of libmgx).

Per rank the results of a batch are two device buffers (mgx_device_results): `n_reads` fixed-size 64-byte headers
(ReadResult, align_types.hpp) and a stream of 32-bit words holding each alignment's node ids, CIGAR runs and path
spelling at `header.stream_off`.  The stream's length differs between ranks, so the gather is two-phase:

  1. all_gather of the used word counts (8 bytes per rank);
  2. `gather` of the headers, and `gather` of the first max(used) words of every rank's stream (every rank's stream
     buffer is at least that long: capacity is a function of the batch shape, which is the same on all ranks, and
     the words past `used` are never read).

Rank 0 ends with headers[r] and stream[r] per rank; read i of rank r is alignment record
`stream[r][headers[r][i].stream_off : ...]`, which `mgx_results_from_raw` (C-ABI, host only) decodes into the
`mgx_results` view `mgx_format_tsv` prints from.  torch.distributed is plumbing: backend "nccl" (= RCCL over xGMI)
on GPUs, "gloo" in the CPU tests.
"""
import ctypes as C

import numpy as np
import torch

from . import capi

HEADER_BYTES = 64


class _DevPtr:
    def __init__(self, ptr, n_bytes):
        self.__cuda_array_interface__ = {"shape": (n_bytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def device_result_tensors(A, dev):
    """-> (headers u8[n * 64], stream u8[capacity words * 4], used words): zero-copy views of libmgx's HBM buffers"""
    L = capi.lib()
    hp, hb, nq, sp, sw = C.c_void_p(), C.c_uint64(), C.c_uint64(), C.c_void_p(), C.c_uint64()
    rc = L.mgx_device_results(A.h, C.byref(hp), C.byref(hb), C.byref(nq), C.byref(sp), C.byref(sw))
    assert rc == 0, L.mgx_last_error()
    assert hb.value == HEADER_BYTES
    cap = L.mgx_device_stream_capacity(A.h)
    hdr = torch.as_tensor(_DevPtr(hp.value, hb.value * nq.value), device=dev) if nq.value else torch.empty(0, dtype=torch.uint8, device=dev)
    stream = torch.as_tensor(_DevPtr(sp.value, cap * 4), device=dev) if cap else torch.empty(0, dtype=torch.uint8, device=dev)
    return hdr, stream, int(sw.value)


def gather_raw(dist, rank, world, hdr, stream, used_words):
    """Two-phase gather of (headers, stream) byte tensors to rank 0.  Returns on rank 0 a list of
    (headers u8 tensor, stream u8 tensor cut to that rank's used words) per rank; None elsewhere."""
    dev = hdr.device
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([used_words], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, mine)
    max_words = int(sizes.max().item())
    assert stream.numel() >= 4 * max_words, "stream buffers are sized from the batch shape and must cover every rank's use"
    part = stream[:4 * max_words]
    hl = [torch.empty_like(hdr) for _ in range(world)] if rank == 0 else None
    sl = [torch.empty_like(part) for _ in range(world)] if rank == 0 else None
    dist.gather(hdr, hl, dst=0)
    dist.gather(part, sl, dst=0)
    if rank != 0:
        return None
    return [(hl[r], sl[r][:4 * int(sizes[r].item())]) for r in range(world)]


def gather_device_results(A, dist, rank, world, dev):
    hdr, stream, used = device_result_tensors(A, dev)
    return gather_raw(dist, rank, world, hdr, stream, used)


class RawResults:
    """Host decode of one rank's raw records through the C-ABI (no GPU needed): keeps the native store alive and exposes
    the mgx_results view."""

    def __init__(self, headers_u8, stream_u8):
        L = capi.lib()
        h = np.ascontiguousarray(headers_u8, dtype=np.uint8)
        s = np.ascontiguousarray(stream_u8, dtype=np.uint8)
        assert h.size % HEADER_BYTES == 0 and s.size % 4 == 0
        self._keep = (h, s)
        self.store = C.c_void_p()
        self.res = capi.Results()
        rc = L.mgx_results_from_raw(h.ctypes.data, h.size // HEADER_BYTES, s.ctypes.data, s.size // 4,
                                    C.byref(self.store), C.byref(self.res))
        if rc != 0:
            raise RuntimeError("mgx_results_from_raw: %s" % L.mgx_last_error().decode())

    def close(self):
        if getattr(self, "store", None) and capi is not None:
            capi.lib().mgx_raw_store_free(self.store)
            self.store = None

    __del__ = close

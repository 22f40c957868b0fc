"""Gather of complete alignments to rank 0 (the reference's "gather" is the whole output line of every read,
cli/align.cpp:469-473; here: one process per GPU, results in the device layout of libmgx).

Per rank the results of a batch are two device buffers (mgx_device_results): `n_reads` fixed-size 64-byte headers
(ReadResult, align_types.hpp) and a stream of 32-bit words holding each alignment's node ids, CIGAR runs and path
spelling at `header.stream_off`.  The stream's length differs between ranks, so the gather is two-phase:

  1. all_gather of the used word counts (8 bytes per rank);
  2. `gather` of the headers, and `gather` of the first max(used) words of every rank's stream (a rank whose buffer is
     shorter than that — capacities are sized per rank — sends a zero-padded temporary; the words past `used` are never
     read).

`ResultGatherer` runs this as a pipeline stage: the collectives are launched asynchronously on a snapshot of the batch's
results (~0.7 KB per read: 7 GB per rank at 10 M reads) and waited for when the next batch has been aligned, so the xGMI
transfer overlaps the next batch's kernels; rank 0's receive buffers are allocated once.

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


def _stream_part(stream, max_words):
    """the first max_words words of a rank's stream buffer; a buffer shorter than that (capacities are sized per rank and a
    rank that re-ran a stage after an overflow has a bigger one than its peers) is padded into a temporary"""
    if stream.numel() >= 4 * max_words:
        return stream[:4 * max_words]
    part = torch.zeros(4 * max_words, dtype=torch.uint8, device=stream.device)
    part[:stream.numel()] = stream
    return part


def gather_raw(dist, rank, world, hdr, stream, used_words):
    """Two-phase gather of (headers, stream) byte tensors to rank 0.  Returns on rank 0 a list of
    (headers u8 tensor, stream u8 tensor cut to that rank's used words) per rank; None elsewhere."""
    g = ResultGatherer(dist, rank, world)
    g.start(hdr, stream, used_words, stage=False)
    return g.finish()


class ResultGatherer:
    """The gather as a pipeline stage: start() snapshots a batch's device results and launches the collectives
    asynchronously (RCCL runs them on its own stream), finish() waits for them — so the gather of batch i travels over xGMI
    while batch i + 1 is being aligned.  Receive buffers on rank 0 are allocated once and reused (they only grow).

    Every rank must call start() / finish() in the same order with equally many reads (the header block is fixed-size)."""

    def __init__(self, dist, rank, world):
        self.dist, self.rank, self.world = dist, rank, world
        self._recv_h = self._recv_s = None
        self._stage_h = self._stage_s = None
        self._pending = None

    def start(self, hdr, stream, used_words, stage=True):
        assert self._pending is None, "finish() the previous gather first (one receive buffer set)"
        dist, rank, world = self.dist, self.rank, self.world
        dev = hdr.device
        # phase 1: how many stream words does every rank send, and do all ranks agree on the header block?
        mine = torch.tensor([used_words, hdr.numel()], dtype=torch.int64, device=dev)
        sizes = torch.zeros(2 * world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(sizes, mine)
        sizes = sizes.view(world, 2).cpu()
        assert int(sizes[:, 1].min()) == int(sizes[:, 1].max()), "ranks must gather equally many reads per batch"
        max_words = int(sizes[:, 0].max())
        part = _stream_part(stream, max_words)
        if stage:
            # snapshot: the aligner's buffers are overwritten by the next batch while the collectives are in flight
            if self._stage_h is None or self._stage_h.numel() != hdr.numel():
                self._stage_h = torch.empty_like(hdr)
            if self._stage_s is None or self._stage_s.numel() < part.numel():
                self._stage_s = torch.empty(part.numel() + part.numel() // 8, dtype=torch.uint8, device=dev)
            self._stage_h.copy_(hdr)
            sp = self._stage_s[:part.numel()]
            sp.copy_(part)
            hdr, part = self._stage_h, sp
        hl = sl = None
        if rank == 0:
            if self._recv_h is None or self._recv_h[0].numel() != hdr.numel():
                self._recv_h = [torch.empty_like(hdr) for _ in range(world)]
            if self._recv_s is None or self._recv_s[0].numel() < part.numel():
                cap = part.numel() + part.numel() // 8
                self._recv_s = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
            hl = self._recv_h
            sl = [b[:part.numel()] for b in self._recv_s]
        works = [dist.gather(hdr, hl, dst=0, async_op=True), dist.gather(part, sl, dst=0, async_op=True)]
        self._pending = (works, hl, sl, sizes[:, 0].clone(), (hdr, part))

    def finish(self):
        """-> on rank 0: [(headers u8, stream u8 cut to that rank's used words)] per rank (views of the receive buffers, valid
        until the next start()); None on the other ranks and when nothing is pending"""
        if self._pending is None:
            return None
        works, hl, sl, sizes, _keep = self._pending
        for wk in works:
            if wk is not None:
                wk.wait()
        # Work.wait() orders the CURRENT torch stream behind the collectives; the consumers of these views are not torch
        # kernels (host decode through raw pointers, libmgx on the legacy null stream), so the stream is drained here
        if works and any(wk is not None for wk in works):
            import torch
            if torch.cuda.is_available():
                torch.cuda.current_stream().synchronize()
        self._pending = None
        if self.rank != 0:
            return None
        return [(hl[r], sl[r][:4 * int(sizes[r])]) for r in range(self.world)]


def gather_device_results(A, dist, rank, world, dev):
    hdr, stream, used = device_result_tensors(A, dev)
    return gather_raw(dist, rank, world, hdr, stream, used)


class RawResults:
    """Host decode of one rank's raw records through the C-ABI (no GPU needed): keeps the native store alive and exposes
    the mgx_results view."""

    def __init__(self, headers_u8, stream_u8, labeled=False):
        """labeled: the records come from a label-aware aligner (every alignment is followed by its label list)"""
        L = capi.lib()
        h = np.ascontiguousarray(headers_u8, dtype=np.uint8)
        s = np.ascontiguousarray(stream_u8, dtype=np.uint8)
        assert h.size % HEADER_BYTES == 0 and s.size % 4 == 0
        self._keep = (h, s)
        self.store = C.c_void_p()
        self.res = capi.Results()
        rc = L.mgx_results_from_raw_labeled(h.ctypes.data, h.size // HEADER_BYTES, s.ctypes.data, s.size // 4, 1 if labeled else 0,
                                            C.byref(self.store), C.byref(self.res))
        if rc != 0:
            raise RuntimeError("mgx_results_from_raw: %s" % L.mgx_last_error().decode())

    def close(self):
        if getattr(self, "store", None) and capi is not None:
            capi.lib().mgx_raw_store_free(self.store)
            self.store = None

    __del__ = close

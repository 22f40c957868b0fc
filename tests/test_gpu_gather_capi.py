"""The C-ABI gather of the device results over RCCL (include/mgx.h mgx_gather_*, csrc/mgx_gather.hip) at world size 1 — all this
box has: the communicator, both phases and the root's decode run for real, rank 0 sends to nobody.  (N > 1 cannot be run here;
the exchange is the one tests/test_dist_gloo.py runs at world size 2 on the CPU through metagraph_amd/gather.py.)"""
import ctypes as C
import os
import subprocess

import pytest

from metagraph_amd import capi
from metagraph_amd.aligner import Aligner, pack_queries
from test_gpu_host_adapter import _dump_mt_graph
from test_gpu_parity import gpu_graph
from test_lane_read import bench_like_world
from test_oracle_kats import HERE

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(HERE)


def test_gather_of_one_rank_returns_what_fetch_returns():
    L = capi.lib()
    g, reads = bench_like_world(5, 3000, genome_len=60000)
    reads = list(reads)
    reads.append(("ACGT" * 10).encode() if isinstance(reads[0], bytes) else "ACGT" * 10)      # (no alignment: a "*" line)
    cfg = capi.config_cli(31)
    G = gpu_graph(g)
    A = Aligner(G, cfg)
    want_res, status = A.align_batch(reads)
    assert all(s == 0 for s in status)
    res = A.fetch()
    want = [A.format_tsv(res, i, "r%d" % i, reads[i]) for i in range(len(reads))]
    # the same batch again, results left on the device, through the gather
    blob, offs = pack_queries(reads)
    assert L.mgx_align_batch_device(A.h, blob, offs.ctypes.data, len(reads), 0) == 0, L.mgx_last_error()
    devs = (C.c_int * 1)(0)
    gh = (C.c_void_p * 1)()
    assert L.mgx_gather_create_local(devs, 1, 0, gh) == 0, L.mgx_last_error()
    try:
        assert L.mgx_gather_world(gh[0]) == 1 and L.mgx_gather_rank(gh[0]) == 0
        for _ in range(2):                          # (the handle is reusable: buffers are kept)
            assert L.mgx_gather_start(gh[0], A.h) == 0, L.mgx_last_error()
            nq, words = (C.c_uint64 * 1)(), (C.c_uint64 * 1)()
            hdr, stream = (C.c_void_p * 1)(), (C.c_void_p * 1)()
            assert L.mgx_gather_finish(gh[0], nq, hdr, stream, words) == 0, L.mgx_last_error()
            assert nq[0] == len(reads) and words[0] > 0
            store, raw = C.c_void_p(), capi.Results()
            assert L.mgx_results_from_raw(hdr[0], nq[0], stream[0], words[0], C.byref(store), C.byref(raw)) == 0, L.mgx_last_error()
            got = [A.format_tsv(raw, i, "r%d" % i, reads[i]) for i in range(len(reads))]
            L.mgx_raw_store_free(store)
            assert got == want
        # misuse is an error, not a hang
        assert L.mgx_gather_finish(gh[0], nq, hdr, stream, words) != 0
    finally:
        L.mgx_gather_destroy(gh[0])
    A.close()


def test_mgx_align_rccl_gather_prints_the_same_lines(tmp_path):
    cli, dump = _dump_mt_graph(tmp_path)
    exe = os.path.join(ROOT, "metagraph_amd", "_build", "mgx_align")
    reads = os.path.join(HERE, "golden", cli["reads_fastq"])
    base = [exe, str(dump), reads, "--align-min-exact-match", "0.0"]
    ref = subprocess.run(base, capture_output=True, text=True, timeout=120)
    assert ref.returncode == 0, ref.stderr
    for extra in (["--devices", "1", "--rccl-gather"], ["--devices", "1", "--rccl-gather", "--query-batch-size", "300"]):
        r = subprocess.run(base + extra, capture_output=True, text=True, timeout=180)
        assert r.returncode == 0, r.stderr
        # (RCCL announces its version on stdout when a communicator is made: result lines are the ones with tabs)
        got = [ln for ln in r.stdout.split("\n") if "\t" in ln]
        assert got == [ln for ln in ref.stdout.split("\n") if "\t" in ln]          # one rank, rounds in input order
    r = subprocess.run(base + ["--devices", "2", "--rccl-gather"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "device" in r.stderr

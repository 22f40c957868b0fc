#!/usr/bin/env python3
"""bench.py — aligned reads/s of the DBGAligner hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (k-mer mapping + seeding + extension, `mgx_align_batch_device`)
over one batch of synthetic reads already resident in HBM.  Workload (BASELINE.json configs[1]):
10 M x 150 bp reads against a ~100 M-node k = 31 graph per GPU; with N > 1 every rank owns a replica
of the graph and its own read shard (weak scaling) and the complete alignments (result headers AND the
variable-length node / CIGAR / path-spelling stream) are gathered to rank 0 over RCCL inside the timed region.

`value` is SURVEY 8(d)'s metric: reads start in pinned host memory and the complete results end there — the read H2D and
the result D2H run on a side stream, double-buffered, so that they overlap the kernels of the neighbouring batches
(all --steps batches are timed, the last D2H included).  `value_device_resident` is the same step with reads and
results left in HBM (what rounds 1-3 reported as `value`).

Prints ONE JSON line (rank 0).  PyTorch is plumbing only (device tensors, streams, torch.distributed).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)      # (8: the fill and drain of the host-side double buffering amortised, ~10 s)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=int(os.environ.get("MGX_BENCH_READS", 10_000_000)))
    ap.add_argument("--genome", type=int, default=int(os.environ.get("MGX_BENCH_GENOME", 98_000_000)))
    ap.add_argument("--snps", type=int, default=int(os.environ.get("MGX_BENCH_SNPS", 200_000)))
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--graph-mode", choices=["basic", "primary"], default="basic",
                    help="primary: the same BOSS table declared PRIMARY and aligned through the CanonicalDBG wrapper (the synthetic "
                         "genome's forward strand holds one k-mer of every pair); BASELINE's metric is quoted on basic")
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("MGX_BENCH_CPU_SAMPLE", 200000)))
    ap.add_argument("--parity-sample", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-steps", type=int, default=-1, help="batches of the PCIe-inclusive pipeline (default: --steps; 0 = skip: "
                    "`value` then falls back to the device-resident rate and says so)")
    ap.add_argument("--cpu-1t-sample", type=int, default=1500, help="reads of the single-thread CPU leg")
    ap.add_argument("--labels", type=int, default=0, help="BASELINE config 3: label-aware alignment (LabeledAligner) against an "
                    "annotation of this many labels, label j = the j-th segment of the genome (+ 15 positions into its neighbours)")
    ap.add_argument("--options", default="", help="kernel-selection options for A/B runs, '+'-separated (mgx_aligner_set_pipeline, "
                    "e.g. lane=0: without the lane-per-read kernel); results never depend on them")
    ap.add_argument("--handles", type=int, default=int(os.environ.get("MGX_BENCH_HANDLES", 1)), help="(measurement) a further leg with this many "
                    "aligner handles on streams of their own, one host thread each, the --steps batches dealt round robin — what "
                    "`metagraph align -p N` does on one device (cli/align.cpp:440-475): the kernels of neighbouring batches overlap")
    args = ap.parse_args()
    if args.host_steps < 0:
        args.host_steps = args.steps

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("MGX_BENCH_FORCE_DIST") == "1":     # the latter: single-rank RCCL run of the N > 1 code path
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if dist:
        dist.barrier()
    from metagraph_amd import aligner, capi, synth
    lib = capi.lib()
    assert lib.mgx_device_count() > local_rank, "libmgx.so sees no HIP device (no CPU fallback exists)"

    # ---------------- workload (identical graph on every rank; reads differ per rank) ----------------
    t0 = time.time()
    genome = synth.random_genome(args.genome, 20240501, dev)
    tensors = [genome[None, :]]
    if args.snps:
        tensors.append(synth.snp_windows(genome, args.snps, args.k, 20240502))
    boss = synth.build_boss(tensors, args.k)
    del tensors
    torch.cuda.synchronize()
    n_edges = boss["n_edges"]
    W, last = boss["W"].contiguous(), boss["last"].contiguous()
    G = aligner.Graph(args.k, (W.data_ptr(), n_edges + 1), (last.data_ptr(), n_edges + 1), boss["F"],
                      device=local_rank, on_device=True, mode=2 if args.graph_mode == "primary" else 0)
    t_graph = time.time() - t0
    AN, anno_pairs, t_anno = None, None, 0.0
    if args.labels:
        # the annotation: map the genome's k-mers to nodes (k_map through the C-ABI, chunks of 60 kbp), label j on the nodes of
        # genome segment j widened by 15 positions (so that nodes at the boundaries carry two labels); the SNP windows' alternative
        # k-mers carry no label.  (row, label) pairs go to mgx_annotation_create_sparse from device memory.
        assert args.graph_mode == "basic", "label-aware alignment: BASIC-mode graphs"
        ta = time.time()
        A0 = aligner.Aligner(G, capi.config_cli(args.k))
        chunk, step_len = 60000, 60000 - (args.k - 1)
        starts = torch.arange(0, args.genome - args.k + 1, step_len, device=dev, dtype=torch.int64)
        node_of = torch.zeros(args.genome - args.k + 1, dtype=torch.int64, device=dev)
        lut_acgt = torch.tensor([ord(c) for c in synth.CHARS], dtype=torch.uint8, device=dev)
        per_call = 256
        for c0 in range(0, len(starts), per_call):
            st_c = starts[c0:c0 + per_call]
            lens = torch.minimum(torch.full_like(st_c, chunk), args.genome - st_c)
            offs_c = torch.zeros(len(st_c) + 1, dtype=torch.int64, device=dev)
            offs_c[1:] = torch.cumsum(lens, 0)
            idx = torch.arange(int(offs_c[-1]), device=dev, dtype=torch.int64)
            which = torch.searchsorted(offs_c[1:], idx, right=True)
            seq_c = lut_acgt[genome[st_c[which] + (idx - offs_c[which])].long()].contiguous()
            m = capi.Mapping()
            rcm = lib.mgx_map_batch(A0.h, C.c_void_p(seq_c.data_ptr()), C.c_void_p(offs_c.data_ptr()), len(st_c), 1, C.byref(m))
            assert rcm == 0, lib.mgx_last_error()
            nb = np.ctypeslib.as_array(m.node_begin, shape=(len(st_c) + 1,))
            nf = torch.from_numpy(np.ctypeslib.as_array(m.nodes_fwd, shape=(int(nb[-1]),)).astype(np.int64)).to(dev)
            nbt = torch.from_numpy(nb.astype(np.int64)).to(dev)
            kidx = torch.arange(int(nb[-1]), device=dev, dtype=torch.int64)
            wc = torch.searchsorted(nbt[1:], kidx, right=True)
            node_of[st_c[wc] + (kidx - nbt[wc])] = nf
        del A0
        assert int((node_of == 0).sum()) == 0, "a k-mer of the genome is missing from the graph"
        seg = (args.genome + args.labels - 1) // args.labels
        pos = torch.arange(args.genome - args.k + 1, device=dev, dtype=torch.int64)
        keys = []
        for shift in (0, -15, 15):               # own segment, and the neighbour's within 15 positions of a boundary
            lab = torch.clamp((pos + shift) // seg, 0, args.labels - 1)
            keys.append(lab << 40 | (node_of - 1))
        keys = torch.unique(torch.cat(keys))     # sorted by (label, row), duplicates (a k-mer twice in one segment) removed
        del pos, lab
        labs = keys >> 40
        rows_d = (keys & ((1 << 40) - 1)).contiguous()
        col_begin = torch.zeros(args.labels + 1, dtype=torch.int64, device=dev)
        col_begin[1:] = torch.cumsum(torch.bincount(labs, minlength=args.labels), 0)
        torch.cuda.synchronize()
        AN = aligner.Annotation.from_sparse(n_edges, col_begin.cpu().numpy().astype(np.uint64), rows_d.data_ptr(), device=local_rank, on_device=True)
        anno_pairs = (col_begin.cpu().numpy(), rows_d.cpu().numpy())
        t_anno = time.time() - ta
        del keys, labs, rows_d, node_of
        if rank == 0:
            log("annotation: %d labels, %d (row, label) pairs, %.1f MB on the device, built in %.1fs" %
                (args.labels, len(anno_pairs[1]), AN.device_bytes / 1e6, t_anno))
    reads = synth.sample_reads(genome, args.reads, args.read_len, 20240503 + rank).contiguous()
    offsets = (torch.arange(args.reads + 1, device=dev, dtype=torch.int64) * args.read_len).contiguous()
    torch.cuda.synchronize()
    # the workload generator's temporaries go back to the device: the aligner sizes its per-read arena from free HBM, and
    # torch's caching allocator would otherwise sit on the graph construction's sort buffers for the whole run
    del genome
    torch.cuda.empty_cache()
    if rank == 0:
        log("graph: %d edges, device index %.1f MB, built in %.1fs; reads %d x %d" %
            (n_edges, G.device_bytes / 1e6, t_graph, args.reads, args.read_len))
    cfg = capi.config_cli(args.k)            # `metagraph align` defaults (cli/config/config.hpp:114-145)
    lim = None
    if os.environ.get("MGX_BENCH_LIMITS"):           # tuning probe: "max_columns,cell_arena_bytes" (smaller per-read arena slices)
        mc, cab = [int(v) for v in os.environ["MGX_BENCH_LIMITS"].split(",")]
        lim = capi.Limits()
        lim.max_query_length, lim.max_columns, lim.max_seeds, lim.cell_arena_bytes = 0, mc, 0, cab
    A = aligner.Aligner(G, cfg, lim, annotation=AN)
    for opt in [o for o in args.options.split("+") if o]:
        A.set_pipeline(opt)

    from metagraph_amd import gather as mg

    gatherer = mg.ResultGatherer(dist, rank, world) if dist else None

    def step():
        # Every alignment travels to rank 0 over RCCL/xGMI: fixed-size headers + the variable-length stream (~0.7 KB per read).
        # The gather of batch i is launched asynchronously on a snapshot of the results and travels while batch i + 1 is
        # aligned; it is waited for before the next one starts, and the last one inside the timed region (sync()).
        A.align_device(reads.data_ptr(), offsets.data_ptr(), args.reads)
        if dist:
            gatherer.finish()
            hdr, stream, used = mg.device_result_tensors(A, dev)
            gatherer.start(hdr, stream, used)

    def sync():
        if dist:
            gatherer.finish()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t1 = time.time()
    align_ms, map_ms = [], []
    for _ in range(args.steps):
        step()
        st = A.stats()
        align_ms.append(st["align_kernel_ms"])
        map_ms.append(st["seed_kernel_ms"])
    sync()
    elapsed = time.time() - t1
    if dist:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    st = A.stats()
    ms_per_step = 1000.0 * elapsed / max(1, args.steps)
    total_reads = args.reads * world
    value = total_reads / (elapsed / max(1, args.steps))

    # ---------------- (measurement) several handles at work on the device at once ----------------
    handles_ms = None
    if args.handles > 1 and not dist:
        import ctypes as _ct
        import threading
        L = capi.lib()
        L.mgx_aligner_create_stream.argtypes = [_ct.c_void_p]
        As = [aligner.Aligner(G, cfg, lim, annotation=AN) for _ in range(args.handles)]
        for a in As:
            for opt in [o for o in args.options.split("+") if o]:
                a.set_pipeline(opt)
            assert L.mgx_aligner_create_stream(a.h) == 0
        share = [args.steps // args.handles + (1 if h < args.steps % args.handles else 0) for h in range(args.handles)]

        failed = []

        def work(a, n):
            try:
                for _ in range(n):
                    a.align_device(reads.data_ptr(), offsets.data_ptr(), args.reads)
            except Exception as e:                     # (a thread's exception would otherwise end the thread and nothing else)
                failed.append(e)

        def run(counts):
            ts = [threading.Thread(target=work, args=(a, n)) for a, n in zip(As, counts)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            torch.cuda.synchronize()
            if failed:
                raise failed[0]

        run([1] * args.handles)
        th = time.time()
        run(share)
        handles_ms = 1000.0 * (time.time() - th) / args.steps
        log("handles=%d: %.1f ms per step (one handle: %.1f)" % (args.handles, handles_ms, ms_per_step))
        for a in As:
            a.close()
        del As

    # ---------------- read H2D and result D2H inside the timed region, overlapped with the kernels ----------------
    # Two device read buffers and two pinned result buffers; copies run on a side stream (libmgx's kernels run on the default
    # stream): batch i + 1's reads go up and batch i - 1's results (a device-side snapshot taken right after its kernels, a
    # few ms) come down while batch i is aligned.  The complete results (headers + used part of the stream) reach the host.
    host_value, host_ms = None, None
    if args.host_steps > 0:
        reads_h = reads.cpu().pin_memory()
        offsets_h = offsets.cpu().pin_memory()
        rbuf = [reads, torch.empty_like(reads)]
        obuf = [offsets, torch.empty_like(offsets)]
        hdr_d, stream_d, used0 = mg.device_result_tensors(A, dev)
        hdr_h = [torch.empty(hdr_d.numel(), dtype=torch.uint8).pin_memory() for _ in range(2)]
        stream_h = [torch.empty(min(stream_d.numel(), 4 * used0 + (64 << 20)), dtype=torch.uint8).pin_memory() for _ in range(2)]
        side = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)

        def upload(i):
            ev = torch.cuda.Event()
            with torch.cuda.stream(side):
                rbuf[i % 2].copy_(reads_h, non_blocking=True)
                obuf[i % 2].copy_(offsets_h, non_blocking=True)
                ev.record(side)
            return ev

        def host_pipeline(n_steps):
            up = upload(0)
            down_done = [None, None]
            for i in range(n_steps):
                nxt = upload(i + 1) if i + 1 < n_steps else None
                main.wait_event(up)
                A.align_device(rbuf[i % 2].data_ptr(), obuf[i % 2].data_ptr(), args.reads)
                hd, sd, u = mg.device_result_tensors(A, dev)
                assert 4 * u <= stream_h[i % 2].numel(), "pinned result buffer too small"
                snap_h, snap_s = hd.clone(), sd[:4 * u].clone()          # the next batch overwrites the aligner's buffers
                ready = torch.cuda.Event()
                ready.record(main)
                if down_done[i % 2] is not None:
                    down_done[i % 2].synchronize()                       # (its pinned buffer is free again)
                with torch.cuda.stream(side):
                    side.wait_event(ready)
                    hdr_h[i % 2].copy_(snap_h, non_blocking=True)
                    stream_h[i % 2][:4 * u].copy_(snap_s, non_blocking=True)
                    snap_h.record_stream(side); snap_s.record_stream(side)
                    done = torch.cuda.Event()
                    done.record(side)
                down_done[i % 2] = done
                if dist:
                    mg.gather_device_results(A, dist, rank, world, dev)
                up = nxt
            side.synchronize()

        host_pipeline(1)
        sync()
        th = time.time()
        host_pipeline(args.host_steps)
        sync()
        eh = time.time() - th
        if dist:
            tmax = torch.tensor([eh], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            eh = float(tmax.item())
        host_ms = 1000.0 * eh / args.host_steps
        host_value = total_reads / (eh / args.host_steps)
        del reads_h, offsets_h, hdr_h, stream_h, rbuf, obuf

    if rank != 0:
        if dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---------------- roofline of the dominant kernel (HIP events inside libmgx, per launch) ----------
    k_align = float(np.mean(align_ms))
    k_map = float(np.mean(map_ms))
    lines_total = st["n_rank_lines"] + st["n_select_lines"] + st["n_bit_lines"]
    lines_map = st["n_map_lines"]
    lines_seed = st["n_seed_lines"]
    lines_align = lines_total - lines_map
    n_kmers = max(0, args.read_len - args.k + 1)
    # algorithmic bytes: B_read = L + R_out + 64 B x N_lines.  The seeding and extension kernels also
    # read the two node arrays written by the map kernel; seeds travel between them as 12-B records.
    io_map = args.reads * (2 * args.read_len + 2 * 4 * n_kmers)
    io_seed = args.reads * (args.read_len + 2 * 4 * n_kmers + 32) + 12 * st["n_seeds"]
    io_ext = args.reads * (args.read_len + 96 + 2 * 4 * n_kmers + 32) + 12 * st["n_seeds"]
    split = st["extend_ms"] > 0
    lane_ms = st["lane_ms"]
    lines_lane = st["n_lane_lines"]
    if split:
        # k_lane (one lane per read: finishes the simple reads) and k_extend (8 lanes per read: the reads k_lane passed on) share
        # the extension stage; each read's I/O bytes are charged to the kernel that finished it, every read's seeds and node
        # arrays to k_lane (it looks at all of them)
        lane_reads = st["n_lane_reads"]
        if not (st["extend_kernels"] & capi.KERNEL_LANE):
            lane_ms = 0.0
        ext_reads = args.reads - lane_reads if lane_ms > 0 else args.reads
        io_per_read = args.read_len + 96 + 2 * 4 * n_kmers + 32
        kernels = {"k_map": (k_map, 64.0 * lines_map + io_map),
                   "k_seed": (st["seeding_ms"], 64.0 * lines_seed + io_seed),
                   "k_extend": (st["extend_ms"] - lane_ms, 64.0 * (lines_align - lines_seed - lines_lane) + ext_reads * io_per_read
                                + (12 * st["n_seeds"] if lane_ms <= 0 else 0))}
        kernel_ms = {"k_map": round(k_map, 3), "k_seed": round(st["seeding_ms"], 3), "work_sort": round(st["sort_ms"], 3),
                     "k_extend": round(st["extend_ms"] - lane_ms, 3)}
        if st.get("seed_lane_ms", 0) > 0:
            # (the seeding stage is two launches since round 6: the lane-per-read seeder, then the wave-per-read kernel on what it left)
            kernel_ms["k_seed_lane_part_of_k_seed"] = round(st["seed_lane_ms"], 3)
            kernel_ms["reads_seeded_by_k_seed_lane"] = st["n_seed_lane_reads"]
            kernel_ms["reads_k_seed_lane_left_by_reason"] = st["seed_lane_left_reads"]
        if lane_ms > 0 and (st["extend_kernels"] & capi.KERNEL_LANE):
            kernels["k_lane"] = (lane_ms, 64.0 * lines_lane + args.reads * io_per_read + 12 * st["n_seeds"])
            kernel_ms["k_lane"] = round(lane_ms, 3)
            kernel_ms["reads_finished_by_k_lane"] = lane_reads
            kernel_ms["reads_k_lane_passed_on_by_reason"] = st["lane_bail_reads"]
    dom = max(kernels, key=lambda n: kernels[n][0])
    dom_ms, dom_bytes = kernels[dom]
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    # HBM traffic per launch of the dominant kernel from the committed PMC passes (tools/pmc_passes.sh: FETCH_SIZE and
    # WRITE_SIZE in separate rocprofv3 --pmc runs of this command at 1 M reads, corrected as MI355X_MICROARCH.md
    # prescribes): bytes per read x the reads of this launch.  null when the summary is absent.
    traffic, traffic_src, gather = None, None, None
    try:
        here = os.path.dirname(os.path.abspath(__file__))
        # (label-aware runs have PMC passes of their own: other kernels, other seeds per read)
        pmc_names = ("r06_labels_pmc_summary.json", "r05_labels_pmc_summary.json", "r04_labels_pmc_summary.json") if args.labels else ("r06_pmc_summary.json", "r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json")
        pmc_name = next(n for n in pmc_names if os.path.exists(os.path.join(here, "profiles", n)))
        pmc = json.load(open(os.path.join(here, "profiles", pmc_name)))
        per_read = pmc["kernels"].get(dom, {}).get("traffic_bytes_per_read")
        if per_read:
            traffic = round(per_read * args.reads)
            traffic_src = "profiles/%s (%d-read PMC run, scaled per read)" % (pmc_name, pmc["reads_per_launch"])
        # These kernels gather 64-B lines at random: next to the 8 TB/s streaming peak, every kernel is also stated against the
        # MEASURED ceiling of dependent random 64-B line loads (tools/gather_ceiling.hip, profiles/r02_gather_ceiling.json):
        # fabric lines per second (PMC traffic / 64 B) over the DRAM-resident ceiling.
        ceil_name = next(n for n in ("r03_gather_ceiling.json", "r02_gather_ceiling.json") if os.path.exists(os.path.join(here, "profiles", n)))
        ceil = json.load(open(os.path.join(here, "profiles", ceil_name)))
        dram = max(r["chains1"] for r in ceil["sets"][1]["rows"]) * 1e9
        cache = max(r["chains1"] for r in ceil["sets"][0]["rows"]) * 1e9
        counted = {"k_map": lines_map, "k_seed": lines_seed, "k_extend": lines_align - lines_seed - lines_lane, "k_lane": lines_lane}
        gather = {"ceiling_lines_per_s": {"dram_9GB_set": dram, "infinity_cache_104MB_set": cache},
                  "source": "profiles/" + ceil_name,
                  "note": "block_lines = 64-B BOSS index lines the kernel itself counts (exact); fabric_lines = PMC bytes / 64 B "
                          "(all arrays, Infinity-Cache hits included; FETCH_SIZE as calibrated in profiles/r03_pmc_calibration.json: "
                          "64-B line gathers are tallied 1:1)",
                  "kernels": {}}
        for name, (ms, _) in kernels.items():
            if ms <= 0:
                continue
            e = {"block_lines_per_s": round(counted.get(name, 0) / (ms * 1e-3)),
                 "block_lines_frac_of_dram_ceiling": round(counted.get(name, 0) / (ms * 1e-3) / dram, 3)}
            pr = pmc["kernels"].get(name, {}).get("traffic_bytes_per_read")
            if pr:
                e["fabric_lines_per_s"] = round(pr * args.reads / 64.0 / (ms * 1e-3))
                e["fabric_lines_frac_of_dram_ceiling"] = round(e["fabric_lines_per_s"] / dram, 3)
            gather["kernels"][name] = e
    except (OSError, ValueError, KeyError, IndexError):
        pass
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                "random_line_ceiling": gather,
                "algorithmic_bytes": round(dom_bytes),
                "kernel_ms": kernel_ms,
                "algorithmic_GBps": {n: round(b / (ms * 1e-3) / 1e9, 1) if ms > 0 else 0.0 for n, (ms, b) in kernels.items()},
                "lines_per_read": {"k_map": round(lines_map / args.reads, 1), "k_seed": round(lines_seed / args.reads, 1),
                                   "k_extend": round((lines_align - lines_seed - lines_lane) / args.reads, 1),
                                   "k_lane": round(lines_lane / args.reads, 1)},
                "columns_per_read": round(st["n_columns"] / args.reads, 2),
                "columns_per_read_in_k_lane": round(st["n_lane_columns"] / args.reads, 2),
                # per-group timers of k_extend (a group also "spends" the time it waits for the other 7 reads of its
                # wavefront, so these are shares of wavefront time, not of useful work)
                "phase_share": dict(zip(["prepare", "seed_pickup", "extend", "backtrack", "driver", "output"],
                                        [round(c / max(1, sum(st["phase_cycles"][:6])), 3) for c in st["phase_cycles"][:6]])),
                "extend_share": dict(zip(["pop", "general_step", "chain_step"],
                                         [round(c / max(1, st["phase_cycles"][2]), 3) for c in st["extend_cycles"][:3]]))}

    # ---------------- parity + CPU baseline (oracle = checker, never the thing measured) -----------------
    # The timed CPU leg is the restated reference path built -O3 -march=native -DNDEBUG on this host
    # (oracle/Makefile `fast`), parallelised like cli/align.cpp:415-480 (thread pool over read batches, shared
    # read-only graph).  Every read of its sample is also compared with the GPU's result for the same read.
    parity, cpu = None, None
    if args.parity_sample > 0 or not args.no_cpu_baseline:
        import orc
        orc.use_library(orc.build_fast())
        W_h, last_h = W.cpu().numpy(), last.cpu().numpy()
        view = capi.BossView()
        view.k, view.sigma, view.n_edges, view.mode, view.on_device = args.k, 5, n_edges, (2 if args.graph_mode == "primary" else 0), 0
        view.W, view.last = W_h.ctypes.data, last_h.ctypes.data
        Fc = (C.c_uint64 * 5)(*[int(x) for x in boss["F"]])
        view.F = C.cast(Fc, C.POINTER(C.c_uint64))
        og = orc.Graph(orc.L().orc_graph_from_boss(C.byref(view)))
        # threads the CPU leg may really use: the affinity mask and the cgroup CPU quota, not the machine's core count (the
        # GPU boxes show 256 logical CPUs; profiles/r02_cpu_thread_scan.json: the restated path scales linearly to 16 threads,
        # peaks at 64 and loses 25 % when 256 threads are started)
        threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                threads = max(1, min(threads, int(int(q) / int(per))))
        except (OSError, ValueError):
            pass
        threads = min(threads, int(os.environ.get("MGX_BENCH_CPU_THREADS", 64)))
        orc.L().orc_graph_build_first_chars(og.h, threads)      # NodeFirstCache stand-in (one-off, untimed)
        nc = min(args.parity_sample if args.no_cpu_baseline else max(args.cpu_sample, args.parity_sample), args.reads)
        csample = [bytes(r) for r in reads[:nc].cpu().numpy()]
        tc = time.time()
        if args.labels:
            # the oracle's LabeledAligner on the same annotation (column bit vectors filled from the same pairs); single thread
            nc = min(nc, args.parity_sample)
            csample = csample[:nc]
            oanno = orc.Annotation(og, args.labels)
            cb, rw = anno_pairs
            for j in range(args.labels):
                r = rw[int(cb[j]):int(cb[j + 1])].astype(np.int64)
                np.bitwise_or.at(oanno.column_view(j), r >> 6, (np.uint64(1) << (r & 63).astype(np.uint64)))
            threads = 1
            tc = time.time()
            orun = orc.LabeledAlignRun(og, cfg, oanno, csample, validate=False)
        else:
            orun = orc.AlignRun(og, cfg, csample, threads=threads, validate=False)
        dt = time.time() - tc
        # GPU results of the same reads through the C-ABI with host buffers, compared field by field
        blob, offs = aligner.pack_queries(csample)
        gres = capi.Results()
        rc = lib.mgx_align_batch(A.h, blob, offs.ctypes.data, nc, 0, C.byref(gres))
        assert rc == 0, lib.mgx_last_error()
        ores = capi.Results()
        orc.L().orc_results_view(orun.r, C.byref(ores))
        mism = capi.count_result_mismatches(gres, ores)
        if args.labels:
            # ... and every alignment's label list
            olab = orun.labels()
            ai = 0
            for q in range(nc):
                for ls in olab[q]:
                    if ai < gres.aln_begin[nc]:
                        a = gres.alignments[ai]
                        if [int(gres.labels[a.labels_begin + x]) for x in range(a.n_labels)] != [int(x) for x in ls]:
                            mism += 1
                    ai += 1
            if ai != gres.aln_begin[nc]:
                mism += 1
        cap_err = int(sum(1 for i in range(nc) if gres.status[i] != 0))
        parity = {"sample": nc, "mismatches": int(mism), "capacity_errors": cap_err,
                  "full_batch_capacity_errors": int(st["n_capacity_errors"])}
        if args.labels and not args.no_cpu_baseline:
            cpu = {"value": round(nc / dt, 1), "unit": "reads/s", "cores": 1, "kind": "port",
                   "sample": "first %d reads of the same workload, same graph and annotation, the restated LabeledAligner on 1 thread, %.1fs" % (nc, dt),
                   "build": "-O3 -march=native -DNDEBUG"}
        elif not args.no_cpu_baseline:
            n1 = min(args.cpu_1t_sample, nc)
            t1s = time.time()
            orc.AlignRun(og, cfg, csample[:n1], threads=1, validate=False)
            d1 = time.time() - t1s
            model = "unknown"
            try:
                for line in open("/proc/cpuinfo"):
                    if line.startswith("model name"):
                        model = line.split(":", 1)[1].strip()
                        break
            except OSError:
                pass
            one = n1 / d1
            # thread scan up to the quota (MGX_BENCH_THREAD_SCAN overrides the thread counts): how the restated CPU path scales on
            # this host, next to the quota itself and the whole-node extrapolation below
            scan = {}
            scan_list = [int(x) for x in os.environ["MGX_BENCH_THREAD_SCAN"].split(",")] if os.environ.get("MGX_BENCH_THREAD_SCAN") \
                else sorted(set(t for t in (1, 2, 4, 8, 16, 32, 64, threads) if t <= threads))
            if scan_list:
                for th in scan_list:
                    ns = min(nc, max(2000, 400 * th))
                    ts = time.time()
                    orc.AlignRun(og, cfg, csample[:ns], threads=th, validate=False)
                    scan[str(th)] = round(ns / (time.time() - ts), 1)
            cpu = {"value": round(nc / dt, 1), "unit": "reads/s", "cores": threads, "kind": "port",
                   "sample": "first %d reads of the same workload, same graph, %d threads, %.1fs" % (nc, threads, dt),
                   "build": "-O3 -march=native -DNDEBUG", "cpu_model": model,
                   "single_thread": {"value": round(one, 1), "sample": "%d reads, %.1fs" % (n1, d1)},
                   "thread_scaling_efficiency": round((nc / dt) / (one * threads), 3), "thread_scan": scan,
                   "cpu_quota_threads": threads,
                   # north_star asks for "the node's host cores"; the GPU boxes grant this process a CPU quota (cores above),
                   # so the whole-socket figure can only be extrapolated: single-thread rate x physical cores, which the
                   # measured scaling up to the quota (efficiency above) supports as an upper estimate
                   "extrapolated_whole_socket": {"cores": 128, "value": round(one * 128, 1),
                                                 "note": "single-thread rate x 128 physical cores (2 x EPYC 9575F); not measured"}}

    # `value`: SURVEY 8(d) — host buffers in, host buffers out (the pipelined run above); the device-resident rate next to it
    out = {"metric": "aligned reads/sec (150 bp, k=31)", "value": round(host_value if host_value else value, 1), "unit": "reads/s",
           "value_is": "host-inclusive (read H2D + result D2H overlapped with the kernels)" if host_value else "device-resident (--host-steps 0)",
           "value_device_resident": round(value, 1), "ms_per_step_device_resident": round(ms_per_step, 3),
           "n_gpus": world, "steps": args.host_steps if host_value else args.steps, "warmup": args.warmup,
           "ms_per_step": round(host_ms if host_value else ms_per_step, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
           "config": {"workload": "%d synthetic %d bp reads per GPU vs %d-edge k=%d BOSS graph (%.0f Mbp iid genome + %d SNP windows%s), CLI-default scoring" %
                      (args.reads, args.read_len, n_edges, args.k, args.genome / 1e6, args.snps,
                       ("; PRIMARY mode through the CanonicalDBG wrapper" if args.graph_mode == "primary" else "")
                       + ("; label-aware (LabeledAligner) with a %d-label annotation, one label per genome segment" % args.labels if args.labels else "")),
                      "reads_per_gpu": args.reads, "graph_edges": n_edges, "k": args.k, "graph_mode": args.graph_mode, "parallelism": "reads sharded x%d, graph replicated" % world},
           "roofline": roofline, "cpu_baseline": cpu, "parity": parity}
    if handles_ms is not None:
        out["handles_leg"] = {"handles": args.handles, "ms_per_step": round(handles_ms, 3), "value": round(args.reads / (handles_ms * 1e-3), 1),
                              "note": "device-resident reads, --steps batches dealt to the handles' host threads; measurement only"}
    # the whole step against the same peak (north_star's 40 % is a statement about the path, not about one kernel): the
    # algorithmic bytes of every kernel of a step over the step's wall time (the figure `value` is computed from)
    step_ms = out["ms_per_step"]
    step_bytes = sum(b for _, b in kernels.values())
    roofline["step_algorithmic_bytes"] = round(step_bytes)
    roofline["step_achieved"] = round(step_bytes / (step_ms * 1e-3) / 1e9, 2) if step_ms > 0 else 0.0
    roofline["step_frac"] = round(roofline["step_achieved"] / HBM_PEAK_GBS, 5)
    print(json.dumps(out), flush=True)
    # a batch that lost alignments to a capacity limit is not a valid measurement
    assert st["n_capacity_errors"] == 0, "%d reads ended with a capacity status" % st["n_capacity_errors"]
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""BASELINE.json configs[4] on ONE GPU: a pan-genome-scale graph (default ~1 G edges: `--strains` divergent copies of a
49 Mbp genome, k = 31) with sub-k seeding (`--align-min-seed-length 15`), 1 M reads, a sample checked against the oracle.

Exercises what the 100 M-edge bench does not: 32-bit edge ids near a quarter of their range, the index build at 10x the
size, slot / arena sizing with `max_num_seeds_per_locus = 1000` sub-k seed lists, the seed-stream and output-stream
re-runs.  Prints one JSON line (graph size, index bytes, reads/s, capacity errors, parity of the sample).

    python tools/scale_test.py [--strains 24 --strain-snp 0.05 --genome 49000000 --reads 1000000 --sample 20000]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--strains", type=int, default=24)
    ap.add_argument("--strain-snp", type=float, default=0.05)
    ap.add_argument("--genome", type=int, default=49_000_000)
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--sample", type=int, default=20_000)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--min-seed-length", type=int, default=15)
    ap.add_argument("--read-len", type=int, default=150)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    import __graft_entry__ as ge
    ge.build()
    from metagraph_amd import aligner, capi, synth

    t0 = time.time()
    base = synth.random_genome(args.genome, 20240510, dev)
    strains = []
    g = torch.Generator(device=dev)
    for i in range(args.strains):
        g.manual_seed(20240511 + i)
        mut = torch.rand(args.genome, generator=g, device=dev) < args.strain_snp
        alt = (base + torch.randint(1, 4, (args.genome,), generator=g, device=dev, dtype=torch.uint8)) % 4
        strains.append(torch.where(mut, alt, base))
    seqs = torch.stack(strains)                      # (strains, genome) uint8 codes
    del strains
    # BOSS construction of ~1 G edges in pieces: k-mer keys per strain, then one sort/unique (synth.build_boss takes a list
    # of 2-D tensors and concatenates their k-mer keys)
    boss = synth.build_boss([seqs[i:i + 1] for i in range(args.strains)], args.k)
    torch.cuda.synchronize()
    t_boss = time.time() - t0
    n_edges = boss["n_edges"]
    W, last = boss["W"].contiguous(), boss["last"].contiguous()
    t1 = time.time()
    G = aligner.Graph(args.k, (W.data_ptr(), n_edges + 1), (last.data_ptr(), n_edges + 1), boss["F"], device=0, on_device=True)
    torch.cuda.synchronize()
    t_index = time.time() - t1
    # reads: uniformly over strains
    per = (args.reads + args.strains - 1) // args.strains
    parts = [synth.sample_reads(seqs[i], per, args.read_len, 20240540 + i) for i in range(args.strains)]
    reads = torch.cat(parts)[:args.reads].contiguous()
    perm = torch.randperm(args.reads, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    reads = reads[perm].contiguous()
    offsets = (torch.arange(args.reads + 1, device=dev, dtype=torch.int64) * args.read_len).contiguous()
    del seqs, parts
    cfg = capi.config_cli(args.k)
    cfg.min_seed_length = args.min_seed_length          # --align-min-seed-length 15 (cli/align.cpp:39-40)
    A = aligner.Aligner(G, cfg)
    A.align_device(reads.data_ptr(), offsets.data_ptr(), args.reads)      # warm-up (allocations, stream sizing re-runs)
    torch.cuda.synchronize()
    t2 = time.time()
    A.align_device(reads.data_ptr(), offsets.data_ptr(), args.reads)
    torch.cuda.synchronize()
    dt = time.time() - t2
    st = A.stats()
    out = {"graph_edges": int(n_edges), "strains": args.strains, "strain_snp": args.strain_snp, "k": args.k,
           "min_seed_length": args.min_seed_length, "device_index_bytes": int(G.device_bytes),
           "boss_build_s": round(t_boss, 1), "index_build_s": round(t_index, 1), "reads": args.reads,
           "reads_per_s": round(args.reads / dt, 1), "ms": round(1000 * dt, 1),
           "kernel_ms": {"k_map": round(st["seed_kernel_ms"], 1), "k_seed": round(st["seeding_ms"], 1), "k_extend": round(st["extend_ms"], 1)},
           "capacity_errors": int(st["n_capacity_errors"]), "seeds_per_read": round(st["n_seeds"] / args.reads, 2),
           "columns_per_read": round(st["n_columns"] / args.reads, 1),
           "extensions_per_read": round(st["n_extensions"] / args.reads, 2),
           "chain_column_fraction": round(st["n_fast_columns"] / max(1, st["n_columns"]), 3),
           "k_extend_group_time_share": dict(zip(["prepare", "seed_pickup", "extend", "backtrack", "driver", "output"],
                                                 [round(c / max(1, sum(st["phase_cycles"][:6])), 3) for c in st["phase_cycles"][:6]])),
           "driver_detail_share": {"seedref+filter_nodes": round(st["phase_cycles"][6] / max(1, sum(st["phase_cycles"][:6])), 3),
                                   "reverse+aggregate": round(st["phase_cycles"][7] / max(1, sum(st["phase_cycles"][:6])), 3)},
           "extend_share": dict(zip(["pop", "general_step", "chain_step"],
                                    [round(c / max(1, st["phase_cycles"][2]), 3) for c in st["extend_cycles"][:3]]))}
    if args.sample > 0:
        import orc
        orc.use_library(orc.build_fast())
        W_h, last_h = W.cpu().numpy(), last.cpu().numpy()
        view = capi.BossView()
        view.k, view.sigma, view.n_edges, view.mode, view.on_device = args.k, 5, n_edges, 0, 0
        view.W, view.last = W_h.ctypes.data, last_h.ctypes.data
        Fc = (C.c_uint64 * 5)(*[int(x) for x in boss["F"]])
        view.F = C.cast(Fc, C.POINTER(C.c_uint64))
        og = orc.Graph(orc.L().orc_graph_from_boss(C.byref(view)))
        threads = os.cpu_count() or 1
        # no NodeFirstCache stand-in at this size (it is O(n k)): the oracle walks bwd() on demand
        ns = min(args.sample, args.reads)
        sample = [bytes(r) for r in reads[:ns].cpu().numpy()]
        tc = time.time()
        orun = orc.AlignRun(og, cfg, sample, threads=threads, validate=False)
        out["oracle_s"] = round(time.time() - tc, 1)
        blob, offs = aligner.pack_queries(sample)
        gres = capi.Results()
        rc = capi.lib().mgx_align_batch(A.h, blob, offs.ctypes.data, ns, 0, C.byref(gres))
        assert rc == 0, capi.lib().mgx_last_error()
        ores = capi.Results()
        orc.L().orc_results_view(orun.r, C.byref(ores))
        out["parity"] = {"sample": ns, "mismatches": int(capi.count_result_mismatches(gres, ores)),
                         "capacity_errors": int(sum(1 for i in range(ns) if gres.status[i] != 0))}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

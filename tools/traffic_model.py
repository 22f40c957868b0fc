#!/usr/bin/env python3
"""CPU-side traffic model of the extension kernel (analysis tool, test infrastructure only).

Runs the extension phase of the wave program under the TRACED host model (tests/emu `make trace`: every load and store
of the program calls a hook, 8 lanes per read like the product kernel) on a scaled-down copy of the bench workload
(random genome + SNP windows, 150-bp reads with 1 % substitutions, 0.05 % indels, 5 % random reads, k = 31, CLI defaults)
and prints, per array of the per-read arena / graph table / batch stream, the distinct 64-byte lines read and written
while one read is processed — what a cache holding one read's working set would pass on to the fabric.  The GPU's PMC
traffic (profiles/*pmc_summary.json) is the measured counterpart; this model says WHERE the lines come from.

    python tools/traffic_model.py [--reads 300] [--genome 200000] [--seed 1] [--json out.json]
"""
import argparse
import ctypes as C
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=300)
    ap.add_argument("--genome", type=int, default=200000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--mode", type=int, default=0, help="0 BASIC, 2 PRIMARY")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    os.environ["MGX_EMU_TRACE_LIB"] = "1"
    os.environ["MGX_EMU_SPLIT"] = "1"
    os.environ["MGX_EMU_TRACE_EXTEND"] = "1"
    os.environ["MGX_EMU_LDS"] = "1100"          # dynamic LDS per 8-lane group of the product launch at 150 bp
    import emu_drv
    import orc
    from metagraph_amd import capi
    from test_emu_vs_oracle import rand_seq, mutate, rc

    rng = random.Random(args.seed)
    k = args.k
    genome = rand_seq(rng, args.genome)
    seqs = [genome]
    for _ in range(args.genome // 490):                      # the bench: 200 000 SNP windows on 98 Mbp
        p = rng.randrange(k, args.genome - k)
        alt = rng.choice([c for c in "ACGT" if c != genome[p]])
        seqs.append(genome[p - k + 1:p] + alt + genome[p + 1:p + k])
    if args.mode == 2:
        from test_oracle_primary_goldens import primary_contigs
        seqs = primary_contigs(seqs, k, "input")[0]
    g = orc.Graph.build(k, seqs, args.mode, False)
    eg = emu_drv.EmuGraph(g, mode=args.mode)
    reads = []
    for i in range(args.reads):
        if rng.random() < 0.05:
            reads.append(rand_seq(rng, 150))
            continue
        p = rng.randrange(0, args.genome - 160)
        r = mutate(rng, genome[p:p + 150], sub=0.01, ins=0.0005, dele=0.0005)
        if rng.random() < 0.5:
            r = rc(r)
        reads.append(r)
    cfg = capi.config_cli(k)
    e = emu_drv.EmuRun(eg, cfg, reads)
    assert e.error == "", e.error
    got, status = e.results()
    assert all(s == 0 for s in status)
    st = e.stats()
    L = emu_drv.L()
    L.emu_trace_report.restype = C.c_uint64
    L.emu_trace_report.argtypes = [C.c_char_p, C.c_uint64]
    buf = C.create_string_buffer(1 << 16)
    L.emu_trace_report(buf, len(buf))
    rows = [ln.split("\t") for ln in buf.value.decode().strip().split("\n")]
    n = int(rows[0][1])
    out = {"reads": n, "columns_per_read": st["columns"] / n, "extensions_per_read": st["extensions"] / n, "regions": {}}
    tot_r = tot_w = 0
    print("%d reads, %.1f columns and %.2f extensions per read (chain path: %.1f %%)" % (
        n, st["columns"] / n, st["extensions"] / n, 100.0 * st["fast_columns"] / max(1, st["columns"])))
    print("%-18s %10s %10s %12s %12s" % ("array", "rd lines", "wr lines", "rd accesses", "wr accesses"))
    for name, rl, wl, ra, wa in sorted(rows[1:], key=lambda r: -(int(r[1]) + int(r[2]))):
        rl, wl, ra, wa = int(rl), int(wl), int(ra), int(wa)
        if rl + wl == 0:
            continue
        tot_r += rl
        tot_w += wl
        out["regions"][name] = {"rd_lines_per_read": rl / n, "wr_lines_per_read": wl / n, "rd_accesses_per_read": ra / n, "wr_accesses_per_read": wa / n}
        print("%-18s %10.1f %10.1f %12.1f %12.1f" % (name, rl / n, wl / n, ra / n, wa / n))
    print("%-18s %10.1f %10.1f   = %.1f KB per read (64-B lines, read + written)" % ("total", tot_r / n, tot_w / n, (tot_r + tot_w) * 64 / n / 1000.0))
    out["lines_per_read"] = {"read": tot_r / n, "written": tot_w / n}
    out["bytes_per_read"] = (tot_r + tot_w) * 64 / n
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()

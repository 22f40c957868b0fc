"""Experiment: what does lock-step between the 8 reads of a wavefront cost the extension kernel?  The same batch is run
(a) as it is and (b) with every read replicated 8x in adjacent positions (n/8 distinct reads), so that the 8 groups of a
wavefront execute identical control flow.  Equal per-read work in both; the k_extend difference is divergence + idling."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from metagraph_amd import aligner, capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k, L = 31, 150
dev = torch.device("cuda", 0)
genome = synth.random_genome(98_000_000, 20240501, dev)
boss = synth.build_boss([genome[None, :], synth.snp_windows(genome, 200_000, k, 20240502)], k)
W, last = boss["W"].contiguous(), boss["last"].contiguous()
G = aligner.Graph(k, (W.data_ptr(), boss["n_edges"] + 1), (last.data_ptr(), boss["n_edges"] + 1), boss["F"], device=0, on_device=True)
reads = synth.sample_reads(genome, n, L, 20240503).contiguous()
offsets = (torch.arange(n + 1, device=dev, dtype=torch.int64) * L).contiguous()
del genome
torch.cuda.empty_cache()
cfg = capi.config_cli(k)
if os.environ.get("PROBE_NO_SDUST"):           # ablation (different results): what does the complexity filter cost k_seed?
    cfg.seed_complexity_filter = 0
if os.environ.get("PROBE_MAX_SEED_K"):         # the seeding of label-aware alignment (LabeledAligner clamps max_seed_length to k: one seed per k-mer)
    cfg.max_seed_length = k
A = aligner.Aligner(G, cfg)


def run(r, tag):
    for _ in range(2):
        A.align_device(r.data_ptr(), offsets.data_ptr(), n)
        torch.cuda.synchronize()
    st = A.stats()
    print(tag, {"k_map": round(st["seed_kernel_ms"], 1), "k_seed": round(st["seeding_ms"], 1), "k_extend": round(st["extend_ms"], 1),
                "columns_per_read": round(st["n_columns"] / n, 1), "extensions_per_read": round(st["n_extensions"] / n, 2)}, flush=True)
    pc, xc = st["phase_cycles"], st["extend_cycles"]
    tot = max(1, sum(pc[:6]))
    if "seedprobe" in os.environ.get("MGX_LIB_PATH", ""):
        names = ["kmer_masks", "base_seeds", "msl init", "lookups", "bookkeeping", "aggregate", "(sdust)", "(enumerate)"]
        tot_s = max(1, pc[0] + pc[1])
        print("   k_seed wave-time ms-equivalents: prepare %.1f seeding %.1f;" % (st["seeding_ms"] * pc[0] / tot_s, st["seeding_ms"] * pc[1] / tot_s),
              {nm: round(st["seeding_ms"] * c / tot_s, 1) for nm, c in zip(names, xc)}, flush=True)
        return
    if "btprobe" in os.environ.get("MGX_LIB_PATH", ""):
        print("   backtrack sections (ms-equivalents): scan %.1f walk %.1f pop+construct %.1f" % tuple(st["extend_ms"] * c / tot for c in xc[3:6]), flush=True)
    elif "probe" in os.environ.get("MGX_LIB_PATH", ""):
        names = ["band+child", "prefetch+shift", "profile+dp", "ins_end+scan", "conv resolve", "children consume", "slot stores", "conv stores"]
        print("   chain_step sections (ms-equivalents):", {nm: round(st["extend_ms"] * c / tot, 1) for nm, c in zip(names, xc)}, flush=True)
        xc = [0] * 8
    print("   k_extend group-time ms-equivalents: prepare %.1f pickup %.1f extend %.1f (pop %.1f general %.1f chain %.1f) backtrack %.1f "
          "driver %.1f (seedref %.1f reverse+aggregate %.1f) output %.1f" % tuple(st["extend_ms"] * c / tot for c in
          (pc[0], pc[1], pc[2], xc[0], xc[1], xc[2], pc[3], pc[4], pc[6], pc[7], pc[5])), flush=True)


run(reads, "distinct reads      ")
if os.environ.get("PROBE_FIRST_ONLY"):
    sys.exit(0)
rep = reads.view(n, L)[torch.arange(n, device=dev) // 8 * 8].contiguous().view(-1)      # reads 0, 8, 16, ... each 8 times
run(rep, "each read 8x in a row")
# control: the same multiset of reads as (b), shuffled, so that cache effects of duplicates are visible separately
perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
run(rep.view(n, L)[perm].contiguous().view(-1), "8x replicated, shuffled")

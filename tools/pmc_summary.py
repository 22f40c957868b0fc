"""Aggregate the rocprofv3 --pmc CSVs of tools/pmc_passes.sh into profiles/<round>_pmc_summary.json:
per kernel the summed counter values over its dispatches, bytes per read, and the SQ ratios."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
out_name = sys.argv[2] if len(sys.argv) > 2 else "r02_pmc_summary.json"
KERNELS = {"k_map": "k_map_pipe", "k_map_lanes": "k_map_packed", "k_pack_reads": "k_pack_reads", "k_map_bytes": "k_map(", "k_seed": "k_align<1", "k_seed_lane": "k_seed_lane", "k_extend": "k_align_grp8", "k_lane": "k_lane"}      # (k_align_grp8<2> or, for PRIMARY graphs / alternative paths, k_align_grp8_alt<2>)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for path in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        for short, pat in KERNELS.items():
            if pat in r["Kernel_Name"]:
                agg[short][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[short].add((path, r["Dispatch_Id"]))
# FETCH_SIZE calibration (profiles/r03_pmc_calibration.json, tools/fetch_calibrate.hip: kernels that move a KNOWN number of
# bytes): dependent random 64-B line gathers are tallied 1.00 : 1 (0.9996 over a 9 GB set, 0.975 over a 104 MB set, i.e.
# Infinity-Cache hits included), streaming 16-B loads 0.50 : 1 (the x2 of MI355X_MICROARCH.md), stores 1.00 : 1.  The
# aligner's kernels gather 64-B lines, so their fetch bytes are FETCH_SIZE x 1024 x 1.0; the x2 figure is kept beside it.
GATHER_KERNELS = ("k_map", "k_map_lanes", "k_map_bytes", "k_seed", "k_seed_lane", "k_extend", "k_lane")
summary = {"reads_per_launch": reads, "note": "bench.py --reads %d --steps 1 --warmup 0: one launch of each kernel; "
           "FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  fetch_bytes_calibrated = FETCH_SIZE x 1024 x f with f = 1.0 "
           "for the gather kernels (64-B line gathers are tallied 1:1, profiles/r03_pmc_calibration.json) and 2.0 for "
           "streaming kernels (MI355X_MICROARCH.md, HBM section); fetch_bytes_guide_x2 applies the guide's x2 to every kernel "
           "(what rounds 1-2 reported).  Infinity-Cache hits are included in these L2-side counters." % reads, "kernels": {}}
for k, c in agg.items():
    d = {"counters": dict(c), "dispatches": len(disp[k])}
    if "FETCH_SIZE" in c:
        d["fetch_bytes_guide_x2"] = 2 * c["FETCH_SIZE"] * 1024
        d["fetch_bytes_calibrated"] = (1.0 if k in GATHER_KERNELS else 2.0) * c["FETCH_SIZE"] * 1024
        d["fetch_bytes_corrected"] = d["fetch_bytes_calibrated"]
    if "WRITE_SIZE" in c:
        d["write_bytes"] = c["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        d["traffic_bytes_per_read"] = (d["fetch_bytes_calibrated"] + d["write_bytes"]) / reads
        d["traffic_bytes_per_read_guide_x2"] = (d["fetch_bytes_guide_x2"] + d["write_bytes"]) / reads
    if c.get("SQ_WAVE_CYCLES"):
        wc = c["SQ_WAVE_CYCLES"]
        d["sq"] = {"active_inst_frac": c.get("SQ_ACTIVE_INST_ANY", 0) / wc, "wait_any_frac": c.get("SQ_WAIT_ANY", 0) / wc,
                   "wait_inst_frac": c.get("SQ_WAIT_INST_ANY", 0) / wc,
                   "insts_per_read": {n[9:]: c[n] / reads for n in c if n.startswith("SQ_INSTS_")}}
    summary["kernels"][k] = d
json.dump(summary, open(os.path.join(ROOT, "profiles", out_name), "w"), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True)[:3000])

#!/bin/bash
# one PMC pass: where do the waves of the bench kernels wait?  LDS / VMEM occupancy levels (level / insts = mean latency in
# cycles), LDS wait cycles, VALU / LDS / scalar busy cycles.  Summarised by tools/pmc_summary.py like the other passes.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
READS=${1:-1000000}
cd /tmp && export TMPDIR=/tmp
rm -rf $REPO/gpurun_out/pmc_lat
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $REPO/gpurun_out/pmc_lat -o lat -- \
    python $REPO/bench.py --reads $READS --steps 1 --warmup 0 --host-steps 0 --no-cpu-baseline --parity-sample 0 > $REPO/gpurun_out/pmc_lat.json 2> $REPO/gpurun_out/pmc_lat.log
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for p in glob.glob("$REPO/gpurun_out/pmc_lat/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        for short, pat in {"k_map": "k_map(", "k_seed": "k_align<1", "k_extend": "k_align_grp8<2>"}.items():
            if pat in r["Kernel_Name"]:
                agg[short][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in agg.items():
    wc = max(1.0, c["SQ_WAVE_CYCLES"])
    print(k, {n: round(v / wc, 4) for n, v in c.items() if n != "SQ_WAVE_CYCLES"},
          "lds_latency_cycles", round(c["SQ_INST_LEVEL_LDS"] / max(1, c["SQ_INSTS_LDS"]), 1),
          "vmem_latency_cycles", round(c["SQ_INST_LEVEL_VMEM"] / max(1, c["SQ_INSTS_VMEM"]), 1))
PY

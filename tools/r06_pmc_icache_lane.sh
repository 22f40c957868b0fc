#!/bin/bash
# round 6: instruction-cache behaviour of k_lane (92 KB of code, a 62 KB column loop; the instruction cache holds 64 KB) and the other bench kernels
REPO=${GRAFT_REPO_ROOT:-/root/repo}
READS=${1:-1000000}
cd /tmp && export TMPDIR=/tmp
rm -rf $REPO/gpurun_out/pmc_ic; mkdir -p $REPO/gpurun_out
timeout 900 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $REPO/gpurun_out/pmc_ic -o ic -- \
    python $REPO/bench.py --reads $READS --steps 1 --warmup 0 --host-steps 0 --no-cpu-baseline --parity-sample 0 > $REPO/gpurun_out/pmc_ic.json 2> $REPO/gpurun_out/pmc_ic.log
python3 - > $REPO/gpurun_out/r06_pmc_icache.txt <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for p in glob.glob("$REPO/gpurun_out/pmc_ic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        for short, pat in {"k_map": "k_map_pipe", "k_seed_lane": "k_seed_lane", "k_lane": "k_lane(", "k_extend": "k_align_grp8<2>"}.items():
            if pat in r["Kernel_Name"]:
                agg[short][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in agg.items():
    print(k, dict(c), "icache miss rate", round(c["SQC_ICACHE_MISSES"] / max(1, c["SQC_ICACHE_REQ"]), 4),
          "insts per ifetch", round(c["SQ_INSTS"] / max(1, c["SQ_IFETCH"]), 2))
PY
cat $REPO/gpurun_out/r06_pmc_icache.txt; tail -3 $REPO/gpurun_out/pmc_ic.log

#!/bin/bash
# PMC passes for the bench kernels (run on the GPU box through gpurun).  Counters are collected in their own
# runs, one counter group per pass, as /opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass; no trace domains next to --pmc).  Output: gpurun_out/pmc_<pass>/...csv
set -x
REPO=${GRAFT_REPO_ROOT:-/root/repo}
READS=${1:-1000000}; shift; EXTRA="$@"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
    name=$1; shift
    timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $REPO/gpurun_out/pmc_$name -o $name -- \
        python $REPO/bench.py --reads $READS --steps 1 --warmup 0 --host-steps 0 --no-cpu-baseline --parity-sample 0 $EXTRA \
        > $REPO/gpurun_out/pmc_$name.json 2> $REPO/gpurun_out/pmc_$name.log
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
ls $REPO/gpurun_out/pmc_*

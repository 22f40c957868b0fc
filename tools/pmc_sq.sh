#!/bin/bash
# one PMC pass (SQ counters) over the bench kernels at 1 M reads; see tools/pmc_passes.sh for the full set
REPO=${GRAFT_REPO_ROOT:-/root/repo}
READS=${1:-1000000}
cd /tmp && export TMPDIR=/tmp
rm -rf $REPO/gpurun_out/pmc_sq
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $REPO/gpurun_out/pmc_sq -o sq -- \
    python $REPO/bench.py --reads $READS --steps 1 --warmup 0 --host-steps 0 --no-cpu-baseline --parity-sample 0 > $REPO/gpurun_out/pmc_sq.json 2> $REPO/gpurun_out/pmc_sq.log

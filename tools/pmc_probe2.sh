#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $REPO/gpurun_out/counters_list.txt 2>&1
run() { name=$1; shift
  rm -rf $REPO/gpurun_out/pmc_$name
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $REPO/gpurun_out/pmc_$name -o $name -- \
    python $REPO/bench.py --reads 1000000 --steps 1 --warmup 0 --host-steps 0 --no-cpu-baseline --parity-sample 0 > $REPO/gpurun_out/pmc_$name.json 2> $REPO/gpurun_out/pmc_$name.log
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS
run b SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM
run c TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum

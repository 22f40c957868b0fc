#!/bin/bash
# PC sampling of the three kernels on a 1 M-read batch (rocprofv3 beta feature): tools/pc_sample.sh [host_trap|stochastic] [interval]
# Writes gpurun_out/pcs_<method>_summary.txt = samples per (kernel, instruction) aggregated by tools/pc_sample_summary.py.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
REPO=$PWD
method=${1:-host_trap}; interval=${2:-1}
unit=time; [ "$method" = stochastic ] && unit=cycles
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pcs_$method
PROBE_FIRST_ONLY=1 timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit \
    --pc-sampling-interval $interval --kernel-trace -d /tmp/pcs_$method -o pcs -- python $REPO/tools/probe_imbalance.py 1000000 \
    > $REPO/gpurun_out/pcs_$method.log 2>&1
echo "rc=$?" >> $REPO/gpurun_out/pcs_$method.log
tail -5 $REPO/gpurun_out/pcs_$method.log
find /tmp/pcs_$method -type f | head -20
for f in $(find /tmp/pcs_$method -name '*pc_sampling*.csv'); do
    ls -la $f; head -3 $f
    python $REPO/tools/pc_sample_summary.py $f $(find /tmp/pcs_$method -name '*kernel_trace.csv' | head -1) > $REPO/gpurun_out/pcs_${method}_summary.txt
    head -c 3000 $REPO/gpurun_out/pcs_${method}_summary.txt
done

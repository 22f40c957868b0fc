# round 6 A/B 20 (one box): k_lane compiled with other instruction-scheduling strategies of the AMDGPU back end (-mllvm -amdgpu-sched-strategy=max-ilp
# `_ilp`, iterative-ilp `_iilp`, max-memory-clause `_memc`, -amdgpu-schedule-metric-bias=0 `_bias0`) against the product (`_head`); none spills VGPRs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { MGX_LIB_PATH=metagraph_amd/_build/libmgx$1.so timeout 600 python bench.py --reads 4000000 --steps 3 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_lane','k_extend','reads_finished_by_k_lane') if k in km}, d.get('parity'))"; }
{ for r in 1 2; do for t in _head _ilp _iilp _bias0 _memc; do run $t; done; done; } > gpurun_out/r06_ab20_lane_sched.txt 2>&1
cat gpurun_out/r06_ab20_lane_sched.txt

# round 6 A/B 19 (one box): the last-cell block of the column pass with one level of branching and selects inside (`_flat`: 100 scalar instructions
# fewer per block of four cells) against the product (`_head`); then the lane tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { MGX_LIB_PATH=metagraph_amd/_build/libmgx$1.so timeout 600 python bench.py --reads 4000000 --steps 3 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_lane','k_extend','reads_finished_by_k_lane') if k in km}, d.get('parity'))"; }
{ run _head; run _flat; run _head; run _flat; run _head; run _flat; } > gpurun_out/r06_ab19_lane_flat.txt 2>&1
cat gpurun_out/r06_ab19_lane_flat.txt
timeout 900 python -m pytest tests/test_gpu_lane.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee -a gpurun_out/r06_ab19_lane_flat.txt

# round 6 A/B 17 (one box): a lane's column window begins AT the column's first cell instead of at the multiple of four below it (`_org`:
# three lanes in four save the wavefront a block of four cells per column) against the product with A/B 14 - 16 (`_rp`); then the lane tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { MGX_LIB_PATH=metagraph_amd/_build/libmgx$1.so timeout 600 python bench.py --reads 4000000 --steps 3 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_lane','k_extend','reads_finished_by_k_lane') if k in km}, d.get('parity'))"; }
{ run _rp; run _org; run _rp; run _org; run _rp; run _org; } > gpurun_out/r06_ab17_lane_origin.txt 2>&1
cat gpurun_out/r06_ab17_lane_origin.txt
timeout 900 python -m pytest tests/test_gpu_lane.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee -a gpurun_out/r06_ab17_lane_origin.txt

# round 5 A/B 6: aln_both's loop over the seeds inside the lane kernel (round 4's patch, now with run-time limits): the bench's
# batches keep one seed per read; seed-rich batches (config 5) run the lanes first with no limit, then the multi-pass extension
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_lane.py tests/test_gpu_config5_scaled.py -x -q -m gpu > gpurun_out/r05_ab6_pytest.log 2>&1; tail -2 gpurun_out/r05_ab6_pytest.log
run() { timeout 600 python bench.py --reads 4000000 --steps 4 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 --options $1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_map','k_seed','k_lane','k_extend','reads_finished_by_k_lane')}, d.get('parity'))"; }
{ run lane_rounds=1; run lane_rounds=8+lane_live=2; run lane_rounds=1; 
timeout 1500 python tools/scale_test.py 2> gpurun_out/r05_ab6_scale.log | tail -c 1200; } > gpurun_out/r05_ab6_lane_rounds.txt 2>&1
cat gpurun_out/r05_ab6_lane_rounds.txt

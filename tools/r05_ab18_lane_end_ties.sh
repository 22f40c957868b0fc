# round 5 A/B 18: k_lane resolves the equal-score batches that arise beyond the query's end (the children of a fork there tie level by
# level until the x-drop ends both branches) in its own order instead of passing the read on (bail reason 27: 1.8 % of the bench)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_lane.py -x -q -m gpu > gpurun_out/r05_ab18_pytest.log 2>&1; tail -2 gpurun_out/r05_ab18_pytest.log
run() { MGX_LIB_PATH=$1 timeout 600 python bench.py --reads 4000000 --steps 4 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_map','k_seed','k_lane','k_extend','reads_finished_by_k_lane','reads_k_lane_passed_on_by_reason') if k in km}, d.get('parity'))"; }
B=metagraph_amd/_build
{ for rep in 1 2 3; do run $B/libmgx.so; run $B/libmgx_lanehead.so; done; } > gpurun_out/r05_ab18_lane_end_ties.txt 2>&1
cat gpurun_out/r05_ab18_lane_end_ties.txt

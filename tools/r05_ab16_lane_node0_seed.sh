# round 5 A/B 16: k_lane checks a later seed that ends in the first seed's first node against the merged vector of the replay columns
# (bail reason 7: 0.8 % of the bench's reads) instead of passing the read on
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_lane.py -x -q -m gpu > gpurun_out/r05_ab16_pytest.log 2>&1; tail -2 gpurun_out/r05_ab16_pytest.log
run() { MGX_LIB_PATH=$1 timeout 600 python bench.py --reads 4000000 --steps 4 --no-cpu-baseline --host-steps 0 --cpu-sample 100000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_map','k_seed','k_lane','k_extend','reads_finished_by_k_lane','reads_k_lane_passed_on_by_reason') if k in km}, d.get('parity'))"; }
B=metagraph_amd/_build
{ for rep in 1 2 3; do run $B/libmgx.so; run $B/libmgx_lanehead.so; done; } > gpurun_out/r05_ab16_lane_node0_seed.txt 2>&1
cat gpurun_out/r05_ab16_lane_node0_seed.txt

# round 5 A/B 12: k_lane with fewer memory requests per column — the select rank of a child computed while its block is held (the
# next column skips the line of the node's block), the S row written only for the columns something may read it from
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_lane.py -x -q -m gpu > gpurun_out/r05_ab12_pytest.log 2>&1; tail -2 gpurun_out/r05_ab12_pytest.log
run() { MGX_LIB_PATH=$1 timeout 600 python bench.py --reads 4000000 --steps 4 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_map','k_seed','k_lane','k_extend','reads_finished_by_k_lane') if k in km}, 'lines/read', d['roofline']['lines_per_read'], d.get('parity'))"; }
B=metagraph_amd/_build
{ for rep in 1 2 3; do run $B/libmgx.so; run $B/libmgx_lanehead.so; done; } > gpurun_out/r05_ab12_lane_requests.txt 2>&1
cat gpurun_out/r05_ab12_lane_requests.txt

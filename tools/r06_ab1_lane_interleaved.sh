# round 6 A/B 1: k_lane with wave-interleaved column slots / S rows ([column][word][lane]) against the round-5 library
# (lane-private 18 KB slices).  GPU lane tests first, then three alternating bench pairs at 4 M reads.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_lane.py -x -q -m gpu > gpurun_out/r06_ab1_pytest.log 2>&1; tail -2 gpurun_out/r06_ab1_pytest.log
run() { MGX_LIB_PATH=$1 timeout 600 python bench.py --reads 4000000 --steps 4 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_map','k_seed','k_lane','k_extend','reads_finished_by_k_lane') if k in km}, 'lines/read', d['roofline']['lines_per_read'], d.get('parity'))"; }
B=metagraph_amd/_build
{ for rep in 1 2 3; do run $B/libmgx.so; run $B/libmgx_r05.so; done; } > gpurun_out/r06_ab1_lane_interleaved.txt 2>&1
cat gpurun_out/r06_ab1_lane_interleaved.txt

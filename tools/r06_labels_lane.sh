# round 6: label-aware alignment with the lane-per-read kernel in front of the labeled group kernel (config 3: 1000 labels)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_labels.py tests/test_gpu_lane.py -x -q -m gpu > gpurun_out/r06_labels_lane_pytest.log 2>&1; tail -4 gpurun_out/r06_labels_lane_pytest.log
run() { timeout 900 python bench.py --labels 1000 --reads 4000000 --steps 3 --no-cpu-baseline --host-steps 0 --parity-sample 20000 --options "$1" 2>gpurun_out/r06_labels_lane.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('options[$1]', d['value'], d['ms_per_step'], km, d.get('parity'))"; }
{ run ""; run "lane=0"; run ""; run "lane=0"; } > gpurun_out/r06_labels_lane.txt 2>&1
cat gpurun_out/r06_labels_lane.txt; tail -3 gpurun_out/r06_labels_lane.err

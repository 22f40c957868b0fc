# round 6 A/B 6: the lane-per-read seeder with one seed per k-mer (label-aware batches, config 3: 1000 labels) against the
# seeding kernel alone (option seed_lane=0); GPU tests of the labeled paths first
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_seed_lane.py tests/test_gpu_labels.py tests/test_lane_labels.py -x -q -m gpu > gpurun_out/r06_ab6_pytest.log 2>&1; tail -4 gpurun_out/r06_ab6_pytest.log
run() { timeout 900 python bench.py --labels 1000 --reads 2000000 --steps 3 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 --options "$1" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['value'], d['ms_per_step'], {k: km[k] for k in ('k_map','k_seed','k_seed_lane_part_of_k_seed','reads_seeded_by_k_seed_lane','reads_k_seed_lane_left_by_reason','k_lane','k_extend') if k in km}, d.get('parity'))"; }
{ for rep in 1 2; do run "seed_lane=1"; run "seed_lane=0"; done; } > gpurun_out/r06_ab6_seed_lane_labels.txt 2>&1
cat gpurun_out/r06_ab6_seed_lane_labels.txt

# round 6 A/B 5: the lane-per-read seeder (mgx_seedlane.hip) in front of k_seed against the seeding kernel alone
# (option seed_lane=0).  GPU tests of the new kernel first, then alternating bench pairs at 4 M reads.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
true
run() { timeout 600 python bench.py --reads 4000000 --steps 4 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 --options "$1" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_map','k_seed','k_seed_lane_part_of_k_seed','reads_seeded_by_k_seed_lane','reads_k_seed_lane_left_by_reason','k_lane','k_extend') if k in km}, d.get('parity'))"; }
{ for rep in 1 2; do run "seed_lane=1"; run "seed_lane=2"; run "seed_lane=0"; done; } > gpurun_out/r06_ab5_seed_lane.txt 2>&1
cat gpurun_out/r06_ab5_seed_lane.txt

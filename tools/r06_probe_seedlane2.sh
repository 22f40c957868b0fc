# round 6: the lane-per-read seeder's section timers, first pass only (seed_lane=2) and both passes (seed_lane=1)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { MGX_LIB_PATH=metagraph_amd/_build/libmgx_sl_tm.so MGX_SL_TIMERS=1 timeout 600 python bench.py --reads 4000000 --steps 2 --no-cpu-baseline --host-steps 0 --cpu-sample 0 --options "$1" 2>gpurun_out/err_tm.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_seed','k_seed_lane_part_of_k_seed','reads_seeded_by_k_seed_lane') if k in km})"; grep "k_seed_lane timers" gpurun_out/err_tm.txt | tail -1; }
{ run seed_lane=2; run seed_lane=1; } > gpurun_out/r06_probe_seedlane2.txt 2>&1
cat gpurun_out/r06_probe_seedlane2.txt

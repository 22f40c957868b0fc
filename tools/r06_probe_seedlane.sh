# round 6: where does k_seed_lane spend its time?  Ablation builds (WRONG results, timing only) and the section timers.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { MGX_LIB_PATH=metagraph_amd/_build/libmgx$1.so MGX_SL_TIMERS=1 timeout 600 python bench.py --reads 2000000 --steps 2 --no-cpu-baseline --host-steps 0 --cpu-sample 0 --options "seed_lane=1" 2>gpurun_out/err_$1.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_seed','k_seed_lane_part_of_k_seed','reads_seeded_by_k_seed_lane','reads_k_seed_lane_left_by_reason') if k in km})"; grep "k_seed_lane timers" gpurun_out/err_$1.txt | tail -1; }
{ run ""; run _sl_tm; run _sl_p1; run _sl_p2; run _sl_p16; run _sl_p19; } > gpurun_out/r06_probe_seedlane.txt 2>&1
cat gpurun_out/r06_probe_seedlane.txt

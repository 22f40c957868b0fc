# round 6 A/B 10: k_lane at THREE wavefronts per SIMD for reads of up to 160 characters (the -DMGX_LANE_SHORT build: 13 312 B of LDS,
# 168 VGPRs with spills) against the general build (two wavefronts, 253 VGPRs, no spills); lane_short=0 / 1 on the same library
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { timeout 600 python bench.py --reads 4000000 --steps 3 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 --options "$1" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_map','k_seed','k_lane','k_extend','reads_finished_by_k_lane') if k in km}, d.get('parity'))"; }
{ run "lane_short=1"; run "lane_short=0"; run "lane_short=1"; run "lane_short=0"; } > gpurun_out/r06_ab10_lane_three_waves.txt 2>&1
cat gpurun_out/r06_ab10_lane_three_waves.txt
timeout 900 python -m pytest tests/test_gpu_lane.py -x -q 2>&1 | tail -3

# round 5 A/B 9: the short-read seeding kernel at 8 wavefronts per SIMD (64 VGPRs, 52 spilled: scratch traffic) vs 4 (102 VGPRs, no spills,
# every seeding table in LDS)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { timeout 600 python bench.py --reads 4000000 --steps 4 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 --options $1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_map','k_seed','k_lane','k_extend')}, d.get('parity'))"; }
{ for o in seed_wps=8 seed_wps=4 seed_wps=8 seed_wps=4; do run $o; done; } > gpurun_out/r05_ab9_seed_wps.txt 2>&1
cat gpurun_out/r05_ab9_seed_wps.txt

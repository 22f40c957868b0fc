# round 5 A/B 11: the short-read seeding kernel at 6 / 7 wavefronts per SIMD (80 / 72 VGPRs) against the product's 8 (64 VGPRs, 52 spilled)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { MGX_LIB_PATH=$1 timeout 600 python bench.py --reads 4000000 --steps 4 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_map','k_seed','k_lane','k_extend')}, d.get('parity'))"; }
B=metagraph_amd/_build
{ for rep in 1 2; do run $B/libmgx.so; run $B/libmgx_wps6.so; run $B/libmgx_wps7.so; done; } > gpurun_out/r05_ab11_seed_wps6.txt 2>&1
cat gpurun_out/r05_ab11_seed_wps6.txt

# round 6 A/B 14 (one box): lane_band() as a bit mask built by shift-and-or from the top cell down (`_bm2`: 255 VGPRs, no spills — the
# or-of-selected-bits form of A/B 12 spilled three registers) against the product (`_head`)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { MGX_LIB_PATH=metagraph_amd/_build/libmgx$1.so timeout 600 python bench.py --reads 4000000 --steps 3 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_lane','k_extend','reads_finished_by_k_lane') if k in km}, d.get('parity'))"; }
{ run _head; run _bm2; run _head; run _bm2; run _head; run _bm2; } > gpurun_out/r06_ab14_lane_band2.txt 2>&1
cat gpurun_out/r06_ab14_lane_band2.txt

# round 6 A/B 15 (one box): the S row's packing without the per-cell "too wide" test where m + xdrop <= 127 rules it out (`_s8`: a max and a
# subtraction per cell) against the product with A/B 14's lane_band (`_ref`)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { MGX_LIB_PATH=metagraph_amd/_build/libmgx$1.so timeout 600 python bench.py --reads 4000000 --steps 3 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_lane','k_extend','reads_finished_by_k_lane') if k in km}, d.get('parity'))"; }
{ run _ref; run _s8; run _ref; run _s8; run _ref; run _s8; } > gpurun_out/r06_ab15_lane_s8.txt 2>&1
cat gpurun_out/r06_ab15_lane_s8.txt

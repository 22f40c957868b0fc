# round 6 A/B 13 (one box): the profile's range test as a per-lane bit mask (`_pm`: -3 vector and -1 scalar instruction per cell, 256 VGPRs, no spills)
# against the product of A/B 12 (`_head`)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { MGX_LIB_PATH=metagraph_amd/_build/libmgx$1.so timeout 600 python bench.py --reads 4000000 --steps 3 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_lane','k_extend','reads_finished_by_k_lane') if k in km}, d.get('parity'))"; }
{ run _head; run _pm; run _head; run _pm; run _head; run _pm; } > gpurun_out/r06_ab13_lane_profmask.txt 2>&1
cat gpurun_out/r06_ab13_lane_profmask.txt

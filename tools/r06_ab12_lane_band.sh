# round 6 A/B 12, all on ONE box (A/B 11 compared the diet against numbers of other boxes: box-to-box differences are of the size of the
# effect): `_old` = the lane kernel of the tree before this session (7c22985) linked into today's library, `_mix` = today's lane_read.hpp
# (runs and cold words out of the LDS) with the OLD column pass, `_head` = the product (diet), `_band` = + lane_band() as a bit mask
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { MGX_LIB_PATH=metagraph_amd/_build/libmgx$1.so timeout 600 python bench.py --reads 4000000 --steps 3 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_lane','k_extend','reads_finished_by_k_lane') if k in km}, d.get('parity'))"; }
{ run _old; run _mix; run _head; run _band; run _old; run _mix; run _head; run _band; run _old; run _head; } > gpurun_out/r06_ab12_lane_band.txt 2>&1
cat gpurun_out/r06_ab12_lane_band.txt

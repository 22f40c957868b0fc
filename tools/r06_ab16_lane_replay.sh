# round 6 A/B 16 (one box): the backward pass's replay columns — their character and node requested ahead of the wave-mates' enumeration, the
# "vector kept exactly" flag in LDS instead of a word in HBM (`_rp`) — against the product with A/B 14 and 15 (`_ref2`)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { MGX_LIB_PATH=metagraph_amd/_build/libmgx$1.so timeout 600 python bench.py --reads 4000000 --steps 3 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_lane','k_extend','reads_finished_by_k_lane') if k in km}, d.get('parity'))"; }
{ run _ref2; run _rp; run _ref2; run _rp; run _ref2; run _rp; } > gpurun_out/r06_ab16_lane_replay.txt 2>&1
cat gpurun_out/r06_ab16_lane_replay.txt

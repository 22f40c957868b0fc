# round 6 A/B 22 (one box): k_lane with iterative-ilp (the product, _head) plus: the AMDGPU register-pressure trackers (_trk), no post-RA scheduler (_nopost),
# post-RA scheduling bottom-up (_pbu), trackers without the pass's scheduling fences (_nf), the flattened last-cell block of A/B 19 (_flat: 1 spilled VGPR)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { MGX_LIB_PATH=metagraph_amd/_build/libmgx$1.so timeout 600 python bench.py --reads 4000000 --steps 3 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_lane','k_extend') if k in km}, d.get('parity')['mismatches'])"; }
{ for r in 1 2; do for t in _head _trk _nopost _pbu _nf _flat; do run $t; done; done; } > gpurun_out/r06_ab22_lane_sched2.txt 2>&1
cat gpurun_out/r06_ab22_lane_sched2.txt

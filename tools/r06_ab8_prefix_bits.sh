# round 6 A/B 8: a bitmap over the suffix-range table (one bit per entry: the range is not empty; 128 MB at m = 15) in front of the
# 8.6 GB table — most look-ups of a strand that is not in the graph find nothing.  MGX_PREFIX_BITS=0 builds the graph without it.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { MGX_PREFIX_BITS=$1 timeout 600 python bench.py --reads 4000000 --steps 4 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('bits=$1', d['ms_per_step'], {k: km[k] for k in ('k_map','k_seed','k_lane','k_extend') if k in km}, d['roofline']['lines_per_read'], d.get('parity'))"; }
{ run 1; run 0; run 1; run 0; } > gpurun_out/r06_ab8_prefix_bits.txt 2>&1
cat gpurun_out/r06_ab8_prefix_bits.txt

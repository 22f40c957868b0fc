# round 6: k_lane time against the share of its resident wavefronts that is launched (MGX_LANE_BLOCKS_PCT): latency- or throughput-bound?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { MGX_LANE_BLOCKS_PCT=$1 timeout 600 python bench.py --reads 4000000 --steps 3 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('pct $1', d['ms_per_step'], {k: km[k] for k in ('k_lane','k_extend','reads_finished_by_k_lane') if k in km})"; }
{ for p in 100 75 50 25 100; do run $p; done; } > gpurun_out/r06_probe_pct.txt 2>&1
cat gpurun_out/r06_probe_pct.txt

# round 6: section timers of k_lane (-DMGX_LANE_TIMERS build, MGX_LANE_TIMERS=1 prints them): where a wavefront's cycles go
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
lib=${1:-_tm}; out=${2:-r06_probe_timers}
{ for pct in 100 25; do echo "blocks pct $pct"; MGX_LANE_BLOCKS_PCT=$pct MGX_LANE_TIMERS=1 MGX_LIB_PATH=metagraph_amd/_build/libmgx$lib.so timeout 600 python bench.py --reads 4000000 --steps 2 --warmup 0 --no-cpu-baseline --host-steps 0 --cpu-sample 2000 2>&1 | grep -a "k_lane timers\|kernel_ms" | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print(d['roofline']['kernel_ms'])
    else: print(line.strip())"; done; } > gpurun_out/$out.txt 2>&1
cat gpurun_out/$out.txt

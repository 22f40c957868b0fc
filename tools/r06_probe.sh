# round 6: timing of k_lane variants — usage: bash tools/r06_probe.sh OUTNAME REPS lib1 lib2 ...   (MGX_LIB_PATH A/B at 4 M reads)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=$1; reps=$2; shift; shift
run() { MGX_LIB_PATH=metagraph_amd/_build/libmgx$1.so timeout 600 python bench.py --reads 4000000 --steps 3 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>gpurun_out/$out.err | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_map','k_seed','k_lane','k_extend','reads_finished_by_k_lane') if k in km}, d.get('parity'))
except Exception as e: print('$1', 'failed', e)"; }
{ for rep in $(seq $reps); do for l in "$@"; do run $l; done; done; } > gpurun_out/$out.txt 2>&1
cat gpurun_out/$out.txt

# round 6: k_lane time under an environment switch — usage: bash tools/r06_probe_env.sh OUT VAR v1 v2 ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=$1; var=$2; shift; shift
run() { env $var=$1 timeout 600 python bench.py --reads 4000000 --steps 3 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$var=$1', d['ms_per_step'], {k: km[k] for k in ('k_lane','k_extend','reads_finished_by_k_lane') if k in km}, d['parity']['mismatches'])"; }
{ for v in "$@"; do run $v; done; } > gpurun_out/$out.txt 2>&1
cat gpurun_out/$out.txt

# round 5 A/B 4: k_seed by section (-DMGX_SEED_PROBE build) with the read-tail range handed over by k_map_pipe (MLEN_TAIL)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B=metagraph_amd/_build
{
echo "== default seeder"; PROBE_FIRST_ONLY=1 MGX_LIB_PATH=$B/libmgx_seedprobe.so timeout 300 python tools/probe_imbalance.py 2000000 2>&1 | grep -v "^\s*$\|amdgpu.ids" | tail -3
run() { MGX_LIB_PATH=$1 timeout 600 python bench.py --reads 4000000 --steps 4 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 --options $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1 $2', d['ms_per_step'], {k: km[k] for k in ('k_map','k_seed','k_lane','k_extend')}, 'lines/read', d['roofline']['lines_per_read'], d.get('parity'))"; }
for rep in 1 2; do run $B/libmgx.so map_pipe=1; done
} > gpurun_out/r05_ab4_seed_tail.txt 2>&1
cat gpurun_out/r05_ab4_seed_tail.txt

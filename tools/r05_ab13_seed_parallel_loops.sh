# round 5 A/B 13: k_seed's two sequential position loops (a MEM's cover of msl[], append_suffix_seed's run behind a position) one
# position per lane; section profile of the new build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B=metagraph_amd/_build
{
echo "== sections"; PROBE_FIRST_ONLY=1 MGX_LIB_PATH=$B/libmgx_seedprobe.so timeout 300 python tools/probe_imbalance.py 2000000 2>&1 | grep -v "^\s*$\|amdgpu.ids" | tail -3
run() { MGX_LIB_PATH=$1 timeout 600 python bench.py --reads 4000000 --steps 4 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_map','k_seed','k_lane','k_extend')}, d.get('parity'))"; }
for rep in 1 2 3; do run $B/libmgx.so; run $B/libmgx_seedhead.so; done
} > gpurun_out/r05_ab13_seed_parallel_loops.txt 2>&1
cat gpurun_out/r05_ab13_seed_parallel_loops.txt

# round 6 A/B 21 (one box): the other units compiled with the back end's max-ilp / max-memory-clause scheduling strategies, one unit at a time —
# mgx.hip (k_map_pipe: _mapilp, _mapmemc), mgx_grp.hip (k_align_grp8 = k_extend: _grpilp, _grpmemc), mgx_seedlane.hip (_slmemc) — against the product (_head)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { MGX_LIB_PATH=metagraph_amd/_build/libmgx$1.so timeout 600 python bench.py --reads 4000000 --steps 3 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_map','k_seed','k_lane','k_extend') if k in km}, d.get('parity')['mismatches'])"; }
{ for r in 1 2; do for t in _head _mapilp _mapmemc _grpilp _grpmemc _slmemc; do run $t; done; done; } > gpurun_out/r06_ab21_units_sched.txt 2>&1
cat gpurun_out/r06_ab21_units_sched.txt

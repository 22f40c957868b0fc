# round 6 A/B 11: k_lane's column pass on a diet (lane_column.hpp: no multiplications per cell, E of the cell at hand without its
# range test, saturating add for E[0]'s extension: 509 -> 477 vector instructions per four cells, 13 quarter-rate multiplies fewer)
# = the product library; `_hp`: + the select hint of the next head fetched before the column pass (-DMGX_LANE_HINT_PREFETCH=1)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { MGX_LIB_PATH=metagraph_amd/_build/libmgx$1.so timeout 600 python bench.py --reads 4000000 --steps 3 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_map','k_seed','k_lane','k_extend','reads_finished_by_k_lane') if k in km}, d.get('parity'))"; }
{ run ""; run _hp; run ""; run _hp; } > gpurun_out/r06_ab11_lane_diet.txt 2>&1
cat gpurun_out/r06_ab11_lane_diet.txt
timeout 900 python -m pytest tests/test_gpu_lane.py -x -q 2>&1 | tail -3

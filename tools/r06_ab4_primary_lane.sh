# round 6 A/B 4: the lane-per-read kernel with the PRIMARY / CanonicalDBG branches (libmgx_prim.so) — on the BASIC bench (does the extra code
# cost the common path anything?) and on the PRIMARY bench (how many reads does the lane finish there?)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/r06_probe.sh r06_ab4_basic 2 "" _prim > /dev/null 2>&1
run() { MGX_LIB_PATH=metagraph_amd/_build/libmgx$1.so timeout 900 python bench.py --graph-mode primary --reads 4000000 --steps 2 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('primary $1', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['parity'])"; }
{ cat gpurun_out/r06_ab4_basic.txt; run ""; run _prim; } > gpurun_out/r06_ab4_primary_lane.txt 2>&1
cat gpurun_out/r06_ab4_primary_lane.txt

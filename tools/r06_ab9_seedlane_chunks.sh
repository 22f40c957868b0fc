cd $GRAFT_REPO_ROOT
run() { MGX_LIB_PATH=metagraph_amd/_build/libmgx$1.so timeout 600 python bench.py --reads 4000000 --steps 3 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_seed','k_seed_lane_part_of_k_seed') if k in km}, d.get('parity'))"; }
run ""; run _sl_c4; run _sl_c2; run _sl_w3; run ""

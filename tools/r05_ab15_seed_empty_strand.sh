# round 5 A/B 15: k_seed answers for a strand without matched k-mers in one pass (strand_without_seeds: k_map's match lengths + the
# read-tail walks) instead of running the whole seeder on it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "seed or sub_k or mapping" > gpurun_out/r05_ab15_pytest.log 2>&1; tail -2 gpurun_out/r05_ab15_pytest.log
run() { MGX_LIB_PATH=$1 timeout 600 python bench.py --reads 4000000 --steps 4 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_map','k_seed','k_lane','k_extend')}, 'lines/read', d['roofline']['lines_per_read'], d.get('parity'))"; }
B=metagraph_amd/_build
{ for rep in 1 2 3; do run $B/libmgx.so; run $B/libmgx_seedhead.so; done; } > gpurun_out/r05_ab15_seed_empty_strand.txt 2>&1
cat gpurun_out/r05_ab15_seed_empty_strand.txt

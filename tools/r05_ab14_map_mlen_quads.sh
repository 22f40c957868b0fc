# round 5 A/B 14: k_map_pipe's match-length bytes leave in quads with the node ids (one 4-byte store where a whole quad failed — every
# quad of a strand that is not in the graph — instead of four scattered 1-byte stores)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mapping" > gpurun_out/r05_ab14_pytest.log 2>&1; tail -2 gpurun_out/r05_ab14_pytest.log
run() { MGX_LIB_PATH=$1 timeout 600 python bench.py --reads 4000000 --steps 4 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_map','k_seed','k_lane','k_extend')}, d.get('parity'))"; }
B=metagraph_amd/_build
{ for rep in 1 2 3; do run $B/libmgx.so; run $B/libmgx_maphead.so; done; } > gpurun_out/r05_ab14_map_mlen_quads.txt 2>&1
cat gpurun_out/r05_ab14_map_mlen_quads.txt

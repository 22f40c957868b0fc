# round 5 A/B 2: k_map_pipe with the target-hint table at 3 (product), 4 (10 spilled VGPRs) and 2 wavefronts per SIMD
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mapping or cli_config or kats or goldens or many_reads" > gpurun_out/r05_ab2_pytest.log 2>&1; tail -2 gpurun_out/r05_ab2_pytest.log
run() { MGX_LIB_PATH=$1 timeout 600 python bench.py --reads 4000000 --steps 6 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 --options $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $2', d['ms_per_step'], d['roofline']['kernel_ms']['k_map'], d['roofline']['kernel_ms']['k_seed'], d['roofline']['kernel_ms']['k_lane'], 'lines/read', d['roofline']['lines_per_read'], d.get('parity'))"; }
B=metagraph_amd/_build
{ for rep in 1 2 3; do run $B/libmgx.so map_pipe=1; run $B/libmgx_w3.so map_pipe=1; run $B/libmgx.so map_pipe=0; done; } > gpurun_out/r05_ab2_map_pipe_waves.txt 2>&1
cat gpurun_out/r05_ab2_map_pipe_waves.txt

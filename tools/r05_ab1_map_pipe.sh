# round 5 A/B 1: k_map as the request / response machine (map_pipe.hpp) against the one-step-per-lane machine of rounds 1-4
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mapping or cli_config or kats or goldens or many_reads" > gpurun_out/r05_ab1_pytest.log 2>&1; tail -3 gpurun_out/r05_ab1_pytest.log
for opt in map_pipe=1 map_pipe=0 map_pipe=1 map_pipe=0; do
  timeout 600 python bench.py --reads 4000000 --steps 3 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 --options $opt 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$opt', d['ms_per_step'], d['roofline']['kernel_ms'], 'lines/read', d['roofline']['lines_per_read'], d.get('parity'))" 
done > gpurun_out/r05_ab1_map_pipe.txt 2>&1
cat gpurun_out/r05_ab1_map_pipe.txt

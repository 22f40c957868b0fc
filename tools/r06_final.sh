# round 6, final tree: PMC passes + default bench line + rocprofv3 kernel stats (run_full_bench.sh), the driver's form, config 3
# (1000 labels) with its own PMC passes and CPU leg, the PRIMARY bench, the whole GPU suite
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/run_full_bench.sh r06 > gpurun_out/r06_full_bench.log 2>&1; tail -8 gpurun_out/r06_full_bench.log | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_steps20.json 2> gpurun_out/r06_bench_steps20.log; tail -1 gpurun_out/r06_bench_steps20.json | cut -c1-400
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
bash tools/pmc_passes.sh 1000000 --labels 1000 > gpurun_out/r06_labels_pmc.log 2>&1
python tools/pmc_summary.py 1000000 r06_labels_pmc_summary.json > /dev/null 2>&1; cp profiles/r06_labels_pmc_summary.json gpurun_out/
timeout 1500 python bench.py --labels 1000 --parity-sample 50000 > gpurun_out/r06_bench_labels1000.json 2> gpurun_out/r06_bench_labels1000.log; tail -1 gpurun_out/r06_bench_labels1000.json | cut -c1-500; echo
timeout 900 python bench.py --graph-mode primary --reads 4000000 --steps 3 --cpu-sample 200000 > gpurun_out/r06_primary_bench.json 2> gpurun_out/r06_primary_bench.log; tail -1 gpurun_out/r06_primary_bench.json | cut -c1-300; echo
timeout 1800 python -m pytest tests -q -m gpu -n 4 > gpurun_out/r06_gpu_tests.txt 2>&1; tail -3 gpurun_out/r06_gpu_tests.txt

# round 5, final tree: the whole GPU suite, the default bench line, the driver's form (--steps 20 --warmup 2), rocprofv3 kernel stats, PMC
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -n 4 > gpurun_out/r05_gpu_tests.txt 2>&1; tail -3 gpurun_out/r05_gpu_tests.txt
bash tools/run_full_bench.sh r05 > gpurun_out/r05_full_bench.log 2>&1; tail -6 gpurun_out/r05_full_bench.log | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r05_bench_steps20.json 2> gpurun_out/r05_bench_steps20.log; tail -1 gpurun_out/r05_bench_steps20.json | cut -c1-300

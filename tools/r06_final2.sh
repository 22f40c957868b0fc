# round 6, final tree with the lane-per-read seeder: PMC passes + default bench line + rocprofv3 kernel stats (run_full_bench.sh),
# the driver's form (20 steps)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq gpurun_out/prof
bash tools/run_full_bench.sh r06 > gpurun_out/r06_full_bench.log 2>&1; tail -8 gpurun_out/r06_full_bench.log | cut -c1-400
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_steps20.json 2> gpurun_out/r06_bench_steps20.log; tail -1 gpurun_out/r06_bench_steps20.json | cut -c1-600
cp gpurun_out/prof/*/*kernel_stats.csv gpurun_out/r06_kernel_stats.csv 2>/dev/null || cp gpurun_out/prof/*kernel_stats.csv gpurun_out/r06_kernel_stats.csv
head -12 gpurun_out/r06_kernel_stats.csv | cut -c1-200

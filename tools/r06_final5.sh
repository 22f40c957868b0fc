# round 6, last tree (k_lane compiled with the iterative-ilp scheduling strategy, A/B 20): the GPU suite, PMC passes + default bench line +
# rocprofv3 kernel stats, the driver's form, config 3 with its PMC passes, the 10 M-read property run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r06_gpu_tests_final.txt 2>&1; tail -3 gpurun_out/r06_gpu_tests_final.txt
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq gpurun_out/prof
bash tools/run_full_bench.sh r06 > gpurun_out/r06_full_bench.log 2>&1; tail -8 gpurun_out/r06_full_bench.log | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_steps20.json 2> gpurun_out/r06_bench_steps20.log; tail -1 gpurun_out/r06_bench_steps20.json | cut -c1-400
cp gpurun_out/prof/*/*kernel_stats.csv gpurun_out/r06_kernel_stats.csv 2>/dev/null || cp gpurun_out/prof/*kernel_stats.csv gpurun_out/r06_kernel_stats.csv
head -8 gpurun_out/r06_kernel_stats.csv | cut -c1-160
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
bash tools/pmc_passes.sh 1000000 --labels 1000 > gpurun_out/r06_labels_pmc.log 2>&1
python tools/pmc_summary.py 1000000 r06_labels_pmc_summary.json > /dev/null 2>&1; cp profiles/r06_labels_pmc_summary.json gpurun_out/
timeout 1500 python bench.py --labels 1000 --parity-sample 50000 > gpurun_out/r06_bench_labels1000.json 2> gpurun_out/r06_bench_labels1000.log; tail -1 gpurun_out/r06_bench_labels1000.json | cut -c1-400; echo
{ echo "MGX_PROP_READS=10000000 python -m pytest tests/test_gpu_properties.py -q -m gpu  (round 6, last tree: k_lane with A/B 14 - 16 and the iterative-ilp schedule; BASELINEs full batch)"; MGX_PROP_READS=10000000 timeout 800 python -m pytest tests/test_gpu_properties.py -q -m gpu 2>&1 | tail -2; } > gpurun_out/r06_properties_10M_reads.txt; cat gpurun_out/r06_properties_10M_reads.txt

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/fetch_calibrate.py > gpurun_out/r03_pmc_calibration.json 2> gpurun_out/r03_pmc_calibration.err
cat gpurun_out/r03_pmc_calibration.json | head -60
timeout 300 python tools/latency_one_read.py > gpurun_out/r03_latency_one_read.json 2> gpurun_out/r03_latency.err; cat gpurun_out/r03_latency_one_read.json
MGX_NO_TORCH=1 timeout 300 python -m pytest tests/test_gpu_host_adapter.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
bash tools/run_full_bench.sh r03p --graph-mode primary --reads 4000000 2>&1 | tail -12

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python tools/scale_test.py > gpurun_out/r03_scale_config5_1gpu.json 2> gpurun_out/r03_scale.err; tail -c 1500 gpurun_out/r03_scale_config5_1gpu.json
timeout 600 python tools/config0.py > gpurun_out/r03_config0.json 2> gpurun_out/r03_config0.err; tail -c 800 gpurun_out/r03_config0.json
timeout 900 python -m pytest tests/test_gpu_bench_path.py tests/test_gpu_properties.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4

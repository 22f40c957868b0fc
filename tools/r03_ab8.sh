cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
MGX_NO_TORCH=1 timeout 600 python -m pytest tests/test_gpu_annotation.py tests/test_gpu_config5_scaled.py tests/test_gpu_host_adapter.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8
MGX_NO_TORCH=1 timeout 900 python tools/annotation_bench.py > gpurun_out/r03_annotation_bench.json 2> gpurun_out/r03_annotation_bench.err; cat gpurun_out/r03_annotation_bench.json; tail -3 gpurun_out/r03_annotation_bench.err

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
N=4000000
B=metagraph_amd/_build
run() { tag=$1; shift; echo "== $tag"; env "$@" PROBE_FIRST_ONLY=1 timeout 300 python tools/probe_imbalance.py $N 2>&1 | grep -v "^\s*$\|amdgpu.ids" | head -1; }
{
run default X=1
run k_map_without_its_output_stores MGX_LIB_PATH=$B/libmgx_nostore.so
run k_map_plain_stores MGX_LIB_PATH=$B/libmgx_plainstore.so
MGX_NO_TORCH=1 timeout 600 python -m pytest tests/test_gpu_config5_scaled.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
} > gpurun_out/r03_ab6.txt 2>&1
cat gpurun_out/r03_ab6.txt

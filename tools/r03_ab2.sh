cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
N=2000000
B=metagraph_amd/_build
run() { tag=$1; shift; echo "== $tag"; env "$@" timeout 300 python tools/probe_imbalance.py $N 2>&1 | grep -v "^\s*$\|amdgpu.ids" | tail -8; }
{
$B/dpp_group_test
$B/dpp_test
run dpp_w2 PROBE_FIRST_ONLY=1
run dpp_w3 PROBE_FIRST_ONLY=1 MGX_LIB_PATH=$B/libmgx_w3.so
} > gpurun_out/r03_ab2.txt 2>&1
cat gpurun_out/r03_ab2.txt

# A/B batch 18 (timing ablation, WRONG results): the trace walk without its three stores per step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
N=2000000
B=metagraph_amd/_build
run() { tag=$1; shift; echo "== $tag"; env "$@" timeout 300 python tools/probe_imbalance.py $N 2>&1 | grep -v "^\s*$\|amdgpu.ids" | tail -2; }
{
run product PROBE_FIRST_ONLY=1
run no_walk_stores PROBE_FIRST_ONLY=1 MGX_LIB_PATH=$B/libmgx_nowalkst.so
} > gpurun_out/r03_ab18.txt 2>&1
cat gpurun_out/r03_ab18.txt

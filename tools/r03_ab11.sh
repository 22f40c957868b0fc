# A/B batch 11: eager tail lookup in k_seed, what the complexity filter costs, long-query routing (config 0)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
N=2000000
B=metagraph_amd/_build
run() { tag=$1; shift; echo "== $tag"; env "$@" timeout 300 python tools/probe_imbalance.py $N 2>&1 | grep -v "^\s*$\|amdgpu.ids" | tail -8; }
c0() { tag=$1; shift; echo "== config0 $tag"; env "$@" timeout 300 python tools/config0.py 2>/dev/null | tail -c 600; echo; }
{
run tail_eager PROBE_FIRST_ONLY=1
run no_sdust PROBE_FIRST_ONLY=1 PROBE_NO_SDUST=1
c0 routed X=1
c0 unrouted MGX_LONG_QUERY_BP=0
MGX_NO_TORCH=1 timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "not torch and not torchrun and not batch_order and not properties" 2>&1 | tail -3
} > gpurun_out/r03_ab11.txt 2>&1
cat gpurun_out/r03_ab11.txt

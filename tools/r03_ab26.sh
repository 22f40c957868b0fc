# A/B batch 26: explicit vmcnt(0) in front of the chain step's stores (the wait-count pass then places no vmcnt wait behind them)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { tag=$1; shift; echo "== $tag"; env "$@" timeout 300 python tools/probe_imbalance.py 2000000 2>&1 | grep -v "^\s*$\|amdgpu.ids" | tail -2; }
{
run vmcnt0_before_stores PROBE_FIRST_ONLY=1
MGX_NO_TORCH=1 timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "not torch and not torchrun and not batch_order and not properties" 2>&1 | tail -3
} > gpurun_out/r03_ab26.txt 2>&1
cat gpurun_out/r03_ab26.txt

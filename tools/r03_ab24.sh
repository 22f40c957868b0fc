# A/B batch 24: lane-parallel diagonal runs in the trace walk
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
N=2000000
run() { tag=$1; shift; echo "== $tag"; env "$@" timeout 300 python tools/probe_imbalance.py $N 2>&1 | grep -v "^\s*$\|amdgpu.ids" | tail -2; }
{
run bt_runs PROBE_FIRST_ONLY=1
run step_by_step PROBE_FIRST_ONLY=1 MGX_NO_BT_RUNS=1
MGX_NO_TORCH=1 timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "not torch and not torchrun and not batch_order and not properties" 2>&1 | tail -3
} > gpurun_out/r03_ab24.txt 2>&1
cat gpurun_out/r03_ab24.txt

# A/B batch 27: trace-walk runs with two steps per lane (16 steps per round trip in the 8-lane groups)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { tag=$1; shift; echo "== $tag"; env "$@" timeout 300 python tools/probe_imbalance.py 2000000 2>&1 | grep -v "^\s*$\|amdgpu.ids" | tail -2; }
{
run two_steps_per_lane PROBE_FIRST_ONLY=1
MGX_NO_TORCH=1 timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "not torch and not torchrun and not batch_order and not properties" 2>&1 | tail -3
} > gpurun_out/r03_ab27.txt 2>&1
cat gpurun_out/r03_ab27.txt

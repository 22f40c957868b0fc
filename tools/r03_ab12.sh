# A/B batch 12: rounds with a step budget (parked extensions): bench workload and config 5
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
N=2000000
run() { tag=$1; shift; echo "== $tag"; env "$@" timeout 300 python tools/probe_imbalance.py $N 2>&1 | grep -v "^\s*$\|amdgpu.ids" | tail -8; }
c5() { tag=$1; shift; echo "== config5 $tag"; env "$@" timeout 600 python tools/scale_test.py 2>/dev/null | tail -c 1300; echo; }
{
run budget_off PROBE_FIRST_ONLY=1
run budget_64 PROBE_FIRST_ONLY=1 MGX_EXT_BUDGET=64
run budget_24 PROBE_FIRST_ONLY=1 MGX_EXT_BUDGET=24
c5 budget_32_default X=1
c5 budget_off MGX_EXT_BUDGET=0
c5 budget_12 MGX_EXT_BUDGET=12
} > gpurun_out/r03_ab12.txt 2>&1
cat gpurun_out/r03_ab12.txt

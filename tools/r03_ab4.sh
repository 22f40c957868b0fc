cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
N=4000000
run() { tag=$1; shift; echo "== $tag"; env "$@" PROBE_FIRST_ONLY=1 timeout 300 python tools/probe_imbalance.py $N 2>&1 | grep -v "^\s*$\|amdgpu.ids" | head -1; }
{
run m15_default X=1
run m14 MGX_PREFIX_LEN_MAX=14
run m13 MGX_PREFIX_LEN_MAX=13
run m12 MGX_PREFIX_LEN_MAX=12
run m10 MGX_PREFIX_LEN_MAX=10
} > gpurun_out/r03_ab4.txt 2>&1
cat gpurun_out/r03_ab4.txt

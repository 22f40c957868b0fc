cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
N=2000000
B=metagraph_amd/_build
run() { tag=$1; shift; echo "== $tag"; env "$@" timeout 300 python tools/probe_imbalance.py $N 2>&1 | grep -v "^\s*$\|amdgpu.ids" | tail -8; }
{
MGX_NO_TORCH=1 timeout 400 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "not torch and not torchrun and not batch_order and not transcripts_1000 and not properties" 2>&1 | tail -3
run rounds_full X=1
run legacy_w3 PROBE_FIRST_ONLY=1 MGX_LIB_PATH=$B/libmgx_legacy.so MGX_NO_FLAT=1
} > gpurun_out/r03_ab3.txt 2>&1
cat gpurun_out/r03_ab3.txt

# A/B batch 13: seeding kernel with 1280 B less static LDS (no score rows, sdust interval lists only)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
N=2000000
run() { tag=$1; shift; echo "== $tag"; env "$@" timeout 300 python tools/probe_imbalance.py $N 2>&1 | grep -v "^\s*$\|amdgpu.ids" | tail -8; }
{
run lds_more PROBE_FIRST_ONLY=1 MGX_SEED_LDS_PRINT=1
run lds_more_cap2600 PROBE_FIRST_ONLY=1 MGX_SEED_LDS_CAP=2600
run lds_as_before PROBE_FIRST_ONLY=1 MGX_SEED_LDS_CAP=1952
MGX_NO_TORCH=1 timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "not torch and not torchrun and not batch_order and not properties and not transcripts_1000" 2>&1 | tail -3
} > gpurun_out/r03_ab13.txt 2>&1
cat gpurun_out/r03_ab13.txt

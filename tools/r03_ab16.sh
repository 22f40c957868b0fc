# A/B batch 16: quick pass of the DUST pre-filter (k_seed)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
N=2000000
run() { tag=$1; shift; echo "== $tag"; env "$@" timeout 300 python tools/probe_imbalance.py $N 2>&1 | grep "distinct reads\|k_seed:"; }
{
run quick_pass PROBE_FIRST_ONLY=1 MGX_SEED_LDS_PRINT=1
run no_sdust PROBE_FIRST_ONLY=1 PROBE_NO_SDUST=1
MGX_NO_TORCH=1 timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "not torch and not torchrun and not batch_order and not properties and not transcripts_1000" 2>&1 | tail -3
} > gpurun_out/r03_ab16.txt 2>&1
cat gpurun_out/r03_ab16.txt

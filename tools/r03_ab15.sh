# A/B batch 15: is k_seed bound by per-wavefront latency or by aggregate traffic?  (fewer resident wavefronts)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
N=2000000
run() { tag=$1; shift; echo "== $tag"; env "$@" timeout 300 python tools/probe_imbalance.py $N 2>&1 | grep "distinct reads"; }
{
run waves_100 PROBE_FIRST_ONLY=1
run waves_75 PROBE_FIRST_ONLY=1 MGX_SEED_WAVES_PCT=75
run waves_50 PROBE_FIRST_ONLY=1 MGX_SEED_WAVES_PCT=50
run waves_25 PROBE_FIRST_ONLY=1 MGX_SEED_WAVES_PCT=25
run lds_0 PROBE_FIRST_ONLY=1 MGX_SEED_LDS_CAP=0
run lds_1024 PROBE_FIRST_ONLY=1 MGX_SEED_LDS_CAP=1024
} > gpurun_out/r03_ab15.txt 2>&1
cat gpurun_out/r03_ab15.txt

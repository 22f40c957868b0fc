# A/B batch 25: seeding kernel at 8 / 7 / 6 waves per SIMD on the final tree
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B=metagraph_amd/_build
run() { tag=$1; shift; echo "== $tag"; env "$@" timeout 300 python tools/probe_imbalance.py 2000000 2>&1 | grep "distinct reads\|k_seed:"; }
{
run wps8 PROBE_FIRST_ONLY=1
run wps7 PROBE_FIRST_ONLY=1 MGX_LIB_PATH=$B/libmgx_swps7.so MGX_SEED_LDS_PRINT=1
run wps6 PROBE_FIRST_ONLY=1 MGX_LIB_PATH=$B/libmgx_swps6.so MGX_SEED_LDS_PRINT=1
} > gpurun_out/r03_ab25.txt 2>&1
cat gpurun_out/r03_ab25.txt

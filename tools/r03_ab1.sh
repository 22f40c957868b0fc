# A/B batch 1 of round 3 (one gpurun call): where does k_extend's time go after the traffic cut?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
N=2000000
B=metagraph_amd/_build
run() { tag=$1; shift; echo "== $tag"; env "$@" timeout 300 python tools/probe_imbalance.py $N 2>&1 | grep -v "^\s*$" | tail -8; }
{
run baseline_full X=1
run ablate3_no_conv_no_colstores PROBE_FIRST_ONLY=1 MGX_ABLATE=3
run ablate7_also_no_backtrack PROBE_FIRST_ONLY=1 MGX_ABLATE=7
run half_groups PROBE_FIRST_ONLY=1 MGX_EXT_GROUPS_PCT=50
run quarter_groups PROBE_FIRST_ONLY=1 MGX_EXT_GROUPS_PCT=25
run waves3 PROBE_FIRST_ONLY=1 MGX_LIB_PATH=$B/libmgx_w3.so
run chain_probe PROBE_FIRST_ONLY=1 MGX_LIB_PATH=$B/libmgx_probe.so
run bt_probe PROBE_FIRST_ONLY=1 MGX_LIB_PATH=$B/libmgx_btprobe.so
run no_compact PROBE_FIRST_ONLY=1 MGX_NO_COMPACT=1
run no_alias PROBE_FIRST_ONLY=1 MGX_NO_ALIAS=1
} > gpurun_out/r03_ab1.txt 2>&1
cat gpurun_out/r03_ab1.txt

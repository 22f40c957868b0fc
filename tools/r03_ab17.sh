# A/B batch 17: the DUST quick pass, same box: full test only / quick pass with its rows after rfirst / rows before pos_start
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
N=2000000
B=metagraph_amd/_build
run() { tag=$1; shift; echo "== $tag"; env "$@" timeout 300 python tools/probe_imbalance.py $N 2>&1 | grep "distinct reads"; }
{
for rep in 1 2; do
run noquick PROBE_FIRST_ONLY=1 MGX_LIB_PATH=$B/libmgx_noquick.so
run cur PROBE_FIRST_ONLY=1 MGX_LIB_PATH=$B/libmgx_cur.so
run rowsfirst PROBE_FIRST_ONLY=1 MGX_LIB_PATH=$B/libmgx_rowsfirst.so
run no_sdust PROBE_FIRST_ONLY=1 PROBE_NO_SDUST=1 MGX_LIB_PATH=$B/libmgx_cur.so
done
} > gpurun_out/r03_ab17.txt 2>&1
cat gpurun_out/r03_ab17.txt

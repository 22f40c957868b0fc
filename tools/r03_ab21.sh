# A/B batch 21: config 5 with a separate work key for sub-k seeds (sort order only)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
c5() { tag=$1; shift; echo "== config5 $tag"; env "$@" timeout 600 python tools/scale_test.py 2>/dev/null | tail -c 1100; echo; }
{
c5 subk_key_24 MGX_SUBK_KEY=24
c5 baseline X=1
} > gpurun_out/r03_ab21.txt 2>&1
cat gpurun_out/r03_ab21.txt

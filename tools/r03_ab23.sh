# A/B batch 23: the 64-lane extension kernel for batches spread one read per wavefront
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
echo "== config0, 64 lanes per read"; timeout 300 python tools/config0.py 2>/dev/null | tail -c 900; echo
echo "== config0, 8 lanes per read (MGX_EXT64=0)"; MGX_EXT64=0 timeout 300 python tools/config0.py 2>/dev/null | tail -c 900; echo
timeout 200 python tools/latency_one_read.py 2>/dev/null | tail -1
MGX_NO_TORCH=1 timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "not torch and not torchrun and not batch_order and not properties" 2>&1 | tail -5
} > gpurun_out/r03_ab23.txt 2>&1
cat gpurun_out/r03_ab23.txt

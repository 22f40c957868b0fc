# A/B batch 20: small batches spread over the wavefronts (one read per wavefront instead of 8)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
echo "== config0 spread (default)"; timeout 300 python tools/config0.py 2>/dev/null | tail -c 900; echo
echo "== config0 8 reads per wavefront (as before)"; MGX_GROUPS_PER_WAVE=0 timeout 300 python tools/config0.py 2>/dev/null | tail -c 900; echo
echo "== config0 2 per wavefront"; MGX_GROUPS_PER_WAVE=2 timeout 300 python tools/config0.py 2>/dev/null | tail -c 900; echo
timeout 200 python tools/latency_one_read.py 2>/dev/null | tail -3
MGX_NO_TORCH=1 timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "not torch and not torchrun and not batch_order and not properties" 2>&1 | tail -3
} > gpurun_out/r03_ab20.txt 2>&1
cat gpurun_out/r03_ab20.txt

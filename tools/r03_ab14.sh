# A/B batch 14: PRIMARY graphs on the product-shaped build (3 waves, rounds) vs the alternative-paths build (2 waves)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
MGX_NO_TORCH=1 timeout 600 python -m pytest tests/test_gpu_zz_primary.py tests/test_gpu_canonical.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
echo "== primary bench, _prim build (default)"
timeout 600 python bench.py --graph-mode primary --reads 4000000 --steps 2 --warmup 1 --no-cpu-baseline --host-steps 0 2>&1 | tail -1 | cut -c1-1500
echo "== primary bench, _alt build (rounds 2-3)"
MGX_PRIMARY_ALT_BUILD=1 timeout 600 python bench.py --graph-mode primary --reads 4000000 --steps 2 --warmup 1 --no-cpu-baseline --host-steps 0 2>&1 | tail -1 | cut -c1-1500
} > gpurun_out/r03_ab14.txt 2>&1
cat gpurun_out/r03_ab14.txt

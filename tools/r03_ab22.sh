# A/B batch 22: per-iteration state machine (every group one step of whatever it is doing) vs the rounds, on config 5 and the bench workload
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B=metagraph_amd/_build
{
echo "== bench workload, flat_iteration loop"; PROBE_FIRST_ONLY=1 MGX_LIB_PATH=$B/libmgx_flatiter.so timeout 300 python tools/probe_imbalance.py 2000000 2>&1 | grep -v "^\s*$\|amdgpu.ids" | tail -2
echo "== config5, flat_iteration loop"; MGX_LIB_PATH=$B/libmgx_flatiter.so timeout 600 python tools/scale_test.py 2>/dev/null | tail -c 1100; echo
} > gpurun_out/r03_ab22.txt 2>&1
cat gpurun_out/r03_ab22.txt

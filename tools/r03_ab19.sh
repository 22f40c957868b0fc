# A/B batch 19: concurrency experiment (two handles, per-thread default streams)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B=metagraph_amd/_build
{
echo "== product library (null stream: the two threads serialise)"; timeout 300 python tools/probe_overlap.py 4000000 2>&1 | grep -v amdgpu.ids | tail -4
echo "== per-thread streams, 3 waves per SIMD"; MGX_LIB_PATH=$B/libmgx_pts3.so timeout 300 python tools/probe_overlap.py 4000000 2>&1 | grep -v amdgpu.ids | tail -4
echo "== per-thread streams, 2 waves per SIMD"; MGX_LIB_PATH=$B/libmgx_pts2.so timeout 300 python tools/probe_overlap.py 4000000 2>&1 | grep -v amdgpu.ids | tail -4
echo "== per-thread streams, 2 waves per SIMD, second thread 0.15 s later"; MGX_LIB_PATH=$B/libmgx_pts2.so timeout 300 python tools/probe_overlap.py 4000000 0.15 2>&1 | grep -v amdgpu.ids | tail -4
} > gpurun_out/r03_ab19.txt 2>&1
cat gpurun_out/r03_ab19.txt

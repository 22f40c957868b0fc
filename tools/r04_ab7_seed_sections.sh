# k_seed by section (-DMGX_SEED_PROBE build of mgx.hip): the default seeder (a handful of MEMs per strand) and the seeder of
# label-aware alignment (max_seed_length == k: one seed per matched k-mer), 2 M reads of the bench workload
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B=metagraph_amd/_build
{
echo "== default seeder"; PROBE_FIRST_ONLY=1 MGX_LIB_PATH=$B/libmgx_seedprobe.so timeout 300 python tools/probe_imbalance.py 2000000 2>&1 | grep -v "^\s*$\|amdgpu.ids" | tail -3
echo "== one seed per k-mer (max_seed_length = k)"; PROBE_MAX_SEED_K=1 PROBE_FIRST_ONLY=1 MGX_LIB_PATH=$B/libmgx_seedprobe.so timeout 300 python tools/probe_imbalance.py 2000000 2>&1 | grep -v "^\s*$\|amdgpu.ids" | tail -3
} > gpurun_out/r04_ab7_seed_sections.txt 2>&1
cat gpurun_out/r04_ab7_seed_sections.txt

# A/B batch 10: k_seed sections; config 0 under 2 waves / the legacy per-read program
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
N=2000000
B=metagraph_amd/_build
run() { tag=$1; shift; echo "== $tag"; env "$@" timeout 300 python tools/probe_imbalance.py $N 2>&1 | grep -v "^\s*$\|amdgpu.ids" | tail -8; }
c0() { tag=$1; shift; echo "== config0 $tag"; env "$@" timeout 300 python tools/config0.py 2>/dev/null | tail -c 600; echo; }
{
run seedprobe PROBE_FIRST_ONLY=1 MGX_LIB_PATH=$B/libmgx_seedprobe.so
run chain_probe PROBE_FIRST_ONLY=1 MGX_LIB_PATH=$B/libmgx_probe.so
c0 product X=1
c0 w2 MGX_LIB_PATH=$B/libmgx_w2.so
c0 legacy_w3 MGX_LIB_PATH=$B/libmgx_legacy.so MGX_NO_FLAT=1
c0 nofast MGX_NO_FAST=1
} > gpurun_out/r03_ab10.txt 2>&1
cat gpurun_out/r03_ab10.txt

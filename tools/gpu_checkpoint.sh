# One GPU call per checkpoint of a round: the torch-free GPU parity tests, the default bench line (10 M reads), the PMC passes.
# Usage: gpurun --timeout 1800 -- 'bash tools/gpu_checkpoint.sh TAG [pmc]'
set -x
TAG=${1:-ckpt}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MGX_NO_TORCH=1 timeout 400 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "not torch and not torchrun and not batch_order and not transcripts_1000 and not properties" > gpurun_out/${TAG}_gpu_tests.log 2>&1
tail -3 gpurun_out/${TAG}_gpu_tests.log
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.log
tail -1 gpurun_out/${TAG}_bench.json | cut -c1-1500
if [ "$2" = "pmc" ]; then
    bash tools/pmc_passes.sh 1000000 > gpurun_out/${TAG}_pmc.log 2>&1
    python tools/pmc_summary.py 1000000 ${TAG}_pmc_summary.json > /dev/null 2>&1
    cp profiles/${TAG}_pmc_summary.json gpurun_out/
    python - <<PY
import json
d=json.load(open("profiles/${TAG}_pmc_summary.json"))
for k,v in d["kernels"].items():
    print(k, "traffic B/read", round(v.get("traffic_bytes_per_read",0)), "insts/read", {a:round(b) for a,b in v.get("sq",{}).get("insts_per_read",{}).items()}, "wait", round(v.get("sq",{}).get("wait_any_frac",0),3))
PY
fi

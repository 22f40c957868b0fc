# round evidence on the GPU box: the default bench line, the same command under rocprofv3 --kernel-trace --stats, the PMC
# passes.  Usage: gpurun --timeout 1800 -- 'bash tools/run_full_bench.sh r03 [extra bench args]'
set -x
TAG=${1:-r03}; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
bash tools/pmc_passes.sh 1000000 "$@" > gpurun_out/${TAG}_pmc.log 2>&1
python tools/pmc_summary.py 1000000 ${TAG}_pmc_summary.json > /dev/null 2>&1
cp profiles/${TAG}_pmc_summary.json gpurun_out/
timeout 1200 python bench.py "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.log
tail -1 gpurun_out/${TAG}_bench.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/*
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o ${TAG} -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --host-steps 0 "$@" > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_prof.log
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof/*kernel_trace.csv      # (per-dispatch trace: large; the stats file is what is kept)
ls $GRAFT_REPO_ROOT/gpurun_out/prof | head
cd $GRAFT_REPO_ROOT
python - <<PY
import json
d=json.load(open("profiles/${TAG}_pmc_summary.json"))
for k,v in d["kernels"].items():
    print(k, "traffic B/read", round(v.get("traffic_bytes_per_read",0)), "insts/read", {a:round(b) for a,b in v.get("sq",{}).get("insts_per_read",{}).items()}, "wait", round(v.get("sq",{}).get("wait_any_frac",0),3))
PY

set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
timeout 900 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.log
tail -1 gpurun_out/bench_full.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof | head -20

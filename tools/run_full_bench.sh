# round evidence on the GPU box: the default bench line, then the same command under rocprofv3 --kernel-trace --stats
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
timeout 1200 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.log
tail -1 gpurun_out/bench_full.json | cut -c1-800
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/*
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r02 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --host-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof | head -20
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof/*kernel_trace.csv      # (per-dispatch trace: large; the stats file is what is kept)

# round 5, last tree: the default bench line, the same command under rocprofv3 --kernel-trace --stats, the driver's form (PMC passes and the
# GPU suite: tools/r05_final.sh on the tree before k_pack_reads' SWAR packer, which none of the PMC kernels contains)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof
timeout 600 python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.log; tail -1 gpurun_out/r05_bench.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/*
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r05 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --host-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/r05_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r05_bench_prof.log
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof/*kernel_trace.csv
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r05_bench_steps20.json 2> gpurun_out/r05_bench_steps20.log; tail -1 gpurun_out/r05_bench_steps20.json | cut -c1-300

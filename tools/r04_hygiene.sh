# round-4 evidence hygiene: get_rows with HIP events and device-resident I/O; the PRIMARY bench on the
# _prim build with a 200 k-read parity sample + its rocprofv3 kernel statistics
# usage: gpurun --timeout 1500 -- 'bash tools/r04_hygiene.sh'
set -x
cd $GRAFT_REPO_ROOT
MGX_NO_TORCH=1 timeout 600 python tools/annotation_bench.py > gpurun_out/r04_annotation_get_rows.json 2> gpurun_out/r04_annotation_get_rows.err
cat gpurun_out/r04_annotation_get_rows.json
timeout 900 python bench.py --graph-mode primary --reads 4000000 --cpu-sample 200000 --parity-sample 200000 > gpurun_out/r04_primary_bench.json 2> gpurun_out/r04_primary_bench.err
tail -1 gpurun_out/r04_primary_bench.json | cut -c1-400
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/*
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r04_primary -- python $GRAFT_REPO_ROOT/bench.py --graph-mode primary --reads 4000000 --no-cpu-baseline --host-steps 0 --parity-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/r04_primary_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r04_primary_prof.log
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof/*kernel_trace.csv
head -6 $GRAFT_REPO_ROOT/gpurun_out/prof/r04_primary_kernel_stats.csv | cut -c1-160

# round 5: BASELINE config 3 (label-aware, 1000 labels) and the PRIMARY-graph bench on the final tree
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python bench.py --labels 1000 --parity-sample 50000 --no-cpu-baseline > gpurun_out/r05_bench_labels1000.json 2> gpurun_out/r05_bench_labels1000.log; tail -c 1500 gpurun_out/r05_bench_labels1000.json | cut -c1-400; echo
timeout 900 python bench.py --graph-mode primary --reads 4000000 --steps 3 --cpu-sample 200000 > gpurun_out/r05_primary_bench.json 2> gpurun_out/r05_primary_bench.log; tail -1 gpurun_out/r05_primary_bench.json | cut -c1-400; echo

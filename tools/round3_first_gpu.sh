# First GPU call of the next round (one gpurun call, ~12 min of box time):
#   1. the PRIMARY hardware check without python (seconds), 2. the GPU tests that do not need torch, without torch (seconds:
#   MGX_NO_TORCH=1 skips the minute-long first `import torch` of a fresh box), 3. the first PRIMARY bench line next to a short
#   BASIC one, 4. the kernel stats of the PRIMARY run.
# Usage: gpurun --timeout 1500 -- 'bash tools/round3_first_gpu.sh'   (needs gpurun_in/ from tools/check_primary_on_gpu.sh's inputs)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
[ -d gpurun_in ] && bash tools/check_primary_on_gpu.sh
MGX_NO_TORCH=1 timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -k "not torch and not torchrun and not batch_order and not transcripts_1000" > gpurun_out/r03_gpu_tests_no_torch.log 2>&1
tail -3 gpurun_out/r03_gpu_tests_no_torch.log
timeout 600 python bench.py --graph-mode primary --reads 2000000 --cpu-sample 20000 --cpu-1t-sample 500 > gpurun_out/r03_bench_primary.json 2> gpurun_out/r03_bench_primary.log
tail -1 gpurun_out/r03_bench_primary.json | cut -c1-900
timeout 600 python bench.py --reads 2000000 --no-cpu-baseline --host-steps 0 > gpurun_out/r03_bench_basic_2m.json 2> gpurun_out/r03_bench_basic_2m.log
tail -1 gpurun_out/r03_bench_basic_2m.json | cut -c1-500
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r03_primary -- python $GRAFT_REPO_ROOT/bench.py --graph-mode primary --reads 2000000 --no-cpu-baseline --host-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/r03_bench_primary_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_bench_primary_prof.log
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof/*kernel_trace.csv
ls $GRAFT_REPO_ROOT/gpurun_out/prof | head

# round 5 A/B 7: length of the suffix-range table (MGX_PREFIX_LEN_MAX): does a table that fits the Infinity Cache next to the
# block array (m = 11: 34 MB, m = 12: 134 MB) beat the 8.6 GB one (m = 15) that answers in one line but always from DRAM?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { MGX_PREFIX_LEN_MAX=$1 timeout 600 python bench.py --reads 4000000 --steps 4 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('m=$1', d['ms_per_step'], {k: km[k] for k in ('k_map','k_seed','k_lane','k_extend')}, d['roofline']['lines_per_read'], d.get('parity'))"; }
{ for m in 15 11 12 13 15 11; do run $m; done; } > gpurun_out/r05_ab7_prefix_len.txt 2>&1
cat gpurun_out/r05_ab7_prefix_len.txt

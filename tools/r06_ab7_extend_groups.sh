# round 6 A/B 7: the 8-lane kernel behind k_lane takes 2.7 % of the reads (the ones with several extensions): how many reads per
# wavefront suit them?  groups_per_wave = 8 (default: all), 4, 2, 1 (= the 64-lane kernel, one read per wavefront)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { timeout 600 python bench.py --reads 4000000 --steps 3 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 --options "$1" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print('$1', d['ms_per_step'], {k: km[k] for k in ('k_map','k_seed','k_lane','k_extend') if k in km}, d.get('parity'))"; }
{ run "groups_per_wave=0"; run "groups_per_wave=4"; run "groups_per_wave=2"; run "groups_per_wave=1"; run "groups_per_wave=0"; } > gpurun_out/r06_ab7_extend_groups.txt 2>&1
cat gpurun_out/r06_ab7_extend_groups.txt

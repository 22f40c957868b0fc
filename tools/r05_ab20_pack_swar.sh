# round 5 A/B 20: k_pack_reads packs 32 characters at a time (pack_swar.hpp) instead of a byte loop per strand; k_kmer_counts takes the
# batch's longest read with one atomic per wavefront
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mapping or invalid or lower_case or empty or ragged" > gpurun_out/r05_ab20_pytest.log 2>&1; tail -2 gpurun_out/r05_ab20_pytest.log
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof20; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof20 -o ab20 -- python $GRAFT_REPO_ROOT/bench.py --reads 4000000 --steps 4 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 > $GRAFT_REPO_ROOT/gpurun_out/r05_ab20_bench.json 2>/dev/null
cd $GRAFT_REPO_ROOT; rm -f gpurun_out/prof20/*kernel_trace.csv
python - <<'PY' > gpurun_out/r05_ab20_pack_swar.txt
import csv, json, glob
d=json.loads(open('gpurun_out/r05_ab20_bench.json').read().strip().split('\n')[-1])
print('bench 4 M reads:', d['ms_per_step'], {k: v for k, v in d['roofline']['kernel_ms'].items() if k in ('k_map','k_seed','k_lane','k_extend')}, d.get('parity'))
f=glob.glob('gpurun_out/prof20/*kernel_stats.csv')[0]
for r in list(csv.reader(open(f)))[1:]:
    if r[0].startswith('k_pack_reads') or r[0].startswith('k_map_pipe') or r[0].startswith('k_kmer_counts'): print(r[0][:40], 'calls', r[1], 'avg ms', round(float(r[3])/1e6,3))
PY
cat gpurun_out/r05_ab20_pack_swar.txt

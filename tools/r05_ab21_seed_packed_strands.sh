# round 5 A/B 21: k_seed takes its 2-bit packed strands from k_pack_reads' words instead of encoding the read again (prepare_query)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "seed or sub_k or invalid or lower_case" > gpurun_out/r05_ab21_pytest.log 2>&1; tail -2 gpurun_out/r05_ab21_pytest.log
run() { timeout 600 python bench.py --reads 4000000 --steps 4 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); km=d['roofline']['kernel_ms']; print(d['ms_per_step'], {k: km[k] for k in ('k_map','k_seed','k_lane','k_extend')}, d.get('parity'))"; }
{ run; } > gpurun_out/r05_ab21_seed_packed_strands.txt 2>&1
cat gpurun_out/r05_ab21_seed_packed_strands.txt

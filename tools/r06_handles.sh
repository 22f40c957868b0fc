# round 6: several aligner handles (own streams, one host thread each) at work on the device at once — bench.py --handles H
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { timeout 900 python bench.py --reads $1 --steps 8 --no-cpu-baseline --host-steps 0 --cpu-sample 20000 --handles $2 2>gpurun_out/handles_err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('reads $1 handles $2', 'one handle', d['ms_per_step'], 'handles leg', d.get('handles_leg'), d.get('parity'))"; grep "handles=" gpurun_out/handles_err.txt; }
{ run 10000000 2; run 10000000 3; run 5000000 2; } > gpurun_out/r06_handles.txt 2>&1
cat gpurun_out/r06_handles.txt; tail -5 gpurun_out/handles_err.txt

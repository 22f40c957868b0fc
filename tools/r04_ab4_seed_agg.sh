# A/B of the lane-parallel seed aggregation / ExactSeeder (k_seed): unlabeled 4 M reads and label-aware 1 M reads
cd $GRAFT_REPO_ROOT
MGX_NO_TORCH=1 timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "seed or mapping or cli_config" 2>&1 | tail -2
timeout 300 python bench.py --reads 4000000 --host-steps 0 --no-cpu-baseline --parity-sample 20000 --steps 3 > gpurun_out/ab4_main.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/ab4_main.json')); print('main 4M', d['ms_per_step_device_resident'], d['roofline']['kernel_ms'], d['parity'])"
timeout 300 python bench.py --labels 1000 --reads 1000000 --genome 10000000 --snps 20000 --steps 2 --warmup 1 --host-steps 0 --parity-sample 1000 > gpurun_out/ab4_lab.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/ab4_lab.json')); print('labels 1M', d['value'], d['roofline']['kernel_ms'], d['parity'])"

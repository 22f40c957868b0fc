cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
metagraph_amd/_build/gather_ceiling > gpurun_out/r03_gather_ceiling.json 2> gpurun_out/r03_gather_ceiling.err; cat gpurun_out/r03_gather_ceiling.json
python tools/fetch_calibrate.py > gpurun_out/r03_pmc_calibration.json 2> gpurun_out/r03_pmc_calibration.err
python -c "
import json; d=json.load(open('gpurun_out/r03_pmc_calibration.json'))
for k,v in d['kernels'].items(): print(k, v.get('fetch_counter_bytes_per_known_byte'), v.get('write_counter_bytes_per_known_byte'))"
MGX_NO_TORCH=1 timeout 600 python -m pytest tests/test_gpu_config5_scaled.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -25

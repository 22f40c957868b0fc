cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "" seedold; do
  if [ -n "$v" ]; then export MGX_LIB_PATH=$GRAFT_REPO_ROOT/metagraph_amd/_build/libmgx_$v.so; else unset MGX_LIB_PATH; fi
  timeout 300 python bench.py --reads 4000000 --host-steps 0 --no-cpu-baseline --parity-sample 0 --steps 3 > gpurun_out/ab5_$v.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/ab5_$v.json')); k=d['roofline']['kernel_ms']; print('variant[$v]', d['ms_per_step_device_resident'], k['k_seed'], k['k_map'], k['k_lane'])"
done; done

# A/B of the lane kernel's branch-free last-cell logic (lane_column.hpp): the product library of the tree before the change
# ("") against libmgx_lanenew.so (mgx_lane.o rebuilt from the changed header); same box, alternating, 4 M reads, parity sample on
cd $GRAFT_REPO_ROOT
for v in "" lanenew "" lanenew; do
  if [ -n "$v" ]; then export MGX_LIB_PATH=$GRAFT_REPO_ROOT/metagraph_amd/_build/libmgx_$v.so; else unset MGX_LIB_PATH; fi
  timeout 300 python bench.py --reads 4000000 --host-steps 0 --no-cpu-baseline --parity-sample ${AB_PARITY:-20000} --steps 3 > gpurun_out/ab_$v.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']; print('variant[$v]', d['ms_per_step_device_resident'], k['k_lane'], k['k_extend'], k['reads_finished_by_k_lane'], d.get('parity'))"
done

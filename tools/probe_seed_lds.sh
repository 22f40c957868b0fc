#!/bin/bash
# tuning probe: k_seed time vs the dynamic-LDS bytes it may use per wavefront (which seeding tables sit in LDS)
for cap in ${@:-99999 1792 1200 592 0}; do
  echo "cap $cap: $(MGX_SEED_LDS_CAP=$cap PROBE_FIRST_ONLY=1 timeout 100 python tools/probe_imbalance.py 1000000 2>&1 | grep -m1 "distinct" | cut -c1-200)"
done

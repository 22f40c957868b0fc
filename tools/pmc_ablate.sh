#!/bin/bash
# where does k_extend's fabric traffic come from?  FETCH_SIZE / WRITE_SIZE of the extension kernel with parts of the chain
# step switched off (MGX_ABLATE: 1 = no convergence table, 2 = no column slots, 4 = no backtrack; results are WRONG, only the
# byte counts mean something).  Prints KB per read.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
READS=${1:-1000000}
cd /tmp && export TMPDIR=/tmp
for ab in ${ABLATIONS:-0 1 2 4 7}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_ab
    MGX_ABLATE=$ab timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_ab -o ab -- \
        python $REPO/bench.py --reads $READS --steps 1 --warmup 0 --host-steps 0 --no-cpu-baseline --parity-sample 0 > /dev/null 2> /tmp/pmc_ab.log
    python3 - <<PY
import csv, glob
tot = 0.0
for p in glob.glob("/tmp/pmc_ab/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "k_align_grp8<2>" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"])
mult = 2 if "$c" == "FETCH_SIZE" else 1
print("ablate $ab $c: %.1f KB/read" % (tot * 1024 * mult / $READS / 1000))
PY
  done
done

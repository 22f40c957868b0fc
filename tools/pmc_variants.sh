REPO=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for V in "$@"; do
  export MGX_LIB_PATH=$REPO/metagraph_amd/_build/libmgx_$V.so
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $REPO/gpurun_out/pmcv_${V}_$c -o v -- python $REPO/bench.py --reads 1000000 --steps 1 --warmup 0 --no-cpu-baseline --parity-sample 0 > $REPO/gpurun_out/pmcv_${V}_$c.json 2>/dev/null
  done
  python - <<PY
import csv,glob,json
tot={}
for c in ["FETCH_SIZE","WRITE_SIZE"]:
    for p in glob.glob("$REPO/gpurun_out/pmcv_${V}_%s/**/*counter_collection.csv"%c, recursive=True):
        for r in csv.DictReader(open(p)):
            if "k_align_grp8" in r["Kernel_Name"]: tot[c]=tot.get(c,0)+float(r["Counter_Value"])
d=json.loads(open("$REPO/gpurun_out/pmcv_${V}_WRITE_SIZE.json").read().strip().splitlines()[-1])
print("$V", "k_extend fetch KB/read (x2):", round(2*tot["FETCH_SIZE"]*1024/1e6/1e3,1), "write KB/read:", round(tot["WRITE_SIZE"]*1024/1e6/1e3,1), d["roofline"]["kernel_ms"])
PY
done

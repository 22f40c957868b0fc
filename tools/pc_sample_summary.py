"""Aggregate a rocprofv3 PC-sampling CSV: samples per (kernel of the dispatch, instruction [+ stall columns if present])."""
import collections
import csv
import sys

csv.field_size_limit(1 << 30)
path, ktrace = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None)
disp = {}
if ktrace:
    for r in csv.DictReader(open(ktrace)):
        disp[r.get("Dispatch_Id")] = r.get("Kernel_Name", "?")[:60]
rd = csv.DictReader(open(path))
cols = rd.fieldnames
print("# columns:", cols)
skip = {"Sample_Timestamp", "Exec_Mask", "Correlation_Id", "Dispatch_Id", "Wave_Count", "Wave_In_Group", "Chiplet", "Workgroup_Id_X",
        "Workgroup_Id_Y", "Workgroup_Id_Z", "Hw_Id", "Wave_Id", "Thread_Id"}
keep = [c for c in cols if c not in skip]
agg = collections.Counter()
per_kernel = collections.Counter()
n = 0
for r in rd:
    kname = disp.get(r.get("Dispatch_Id"), r.get("Dispatch_Id"))
    key = (kname,) + tuple(r[c] for c in keep)
    agg[key] += 1
    per_kernel[kname] += 1
    n += 1
print("# samples:", n)
for kname, c in per_kernel.most_common():
    print("# kernel %-60s %d" % (kname, c))
print("# key columns: kernel,", keep)
for key, c in agg.most_common(4000):
    print(c, "\t".join(str(x) for x in key), sep="\t")

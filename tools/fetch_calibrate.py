#!/usr/bin/env python3
"""Calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on known access patterns (tools/fetch_calibrate.hip): run on the GPU box,
    python tools/fetch_calibrate.py > profiles/rNN_pmc_calibration.json
Two separate --pmc passes (no trace domains next to --pmc).  Counters are reported in KiB."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "metagraph_amd", "_build", "fetch_calibrate")
if not os.path.exists(exe):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", exe, os.path.join(ROOT, "tools", "fetch_calibrate.hip")], check=True)
known = None
counters = {}
os.environ["TMPDIR"] = "/tmp"
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    out = os.path.join(ROOT, "gpurun_out", "calib_" + name)
    r = subprocess.run(["rocprofv3", "--pmc", name, "--output-format", "csv", "-d", out, "-o", "calib", "--", exe],
                       capture_output=True, text=True, cwd="/tmp", timeout=600)
    for line in r.stdout.splitlines():
        pass
    js = r.stdout[r.stdout.index("{"):]
    known = json.loads(js)
    for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0]
            if row["Counter_Name"] == name:
                counters.setdefault(k, {}).setdefault(name, 0.0)
                counters[k][name] += float(row["Counter_Value"])
res = {"device": known["device"], "note": "counter KiB x 1024 / known bytes; FETCH_SIZE tallies a 64-B request as ... see ratios", "kernels": {}}
for k, v in known["kernels"].items():
    c = counters.get(k, {})
    e = dict(v)
    e["FETCH_SIZE_KiB"] = c.get("FETCH_SIZE")
    e["WRITE_SIZE_KiB"] = c.get("WRITE_SIZE")
    if v["read_bytes"] and c.get("FETCH_SIZE") is not None:
        e["fetch_counter_bytes_per_known_byte"] = round(c["FETCH_SIZE"] * 1024 / v["read_bytes"], 4)
    if v["written_bytes"] and c.get("WRITE_SIZE") is not None:
        e["write_counter_bytes_per_known_byte"] = round(c["WRITE_SIZE"] * 1024 / v["written_bytes"], 4)
    res["kernels"][k] = e
json.dump(res, sys.stdout, indent=1)
print()

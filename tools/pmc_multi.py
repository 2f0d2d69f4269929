#!/usr/bin/env python
"""Run the fine-kernel workload (tools/pmc_traffic.py) under several separate rocprofv3 --pmc passes and print the
per-dispatch averages of every counter for the fine-level kernels and the k_scale calibration.
usage: pmc_multi.py OUTDIR N [pass ...]   (a pass = counters joined by '+'; environment is inherited)"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

out, n = sys.argv[1], sys.argv[2]
passes = sys.argv[3:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for i, ps in enumerate(passes):
    d = os.path.join(out, "p%d" % i)
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3", "--pmc"] + ps.split("+") + ["--kernel-trace", "--output-format", "csv", "-d", d, "--",
                                                      sys.executable, "tools/pmc_traffic.py", n, n, n]
    subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=400)
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            if "tile" in k or "k_scale" in k:
                acc[k[:36]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for fn in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            if "tile" in k or "k_scale" in k:
                dur[k[:36]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
res = {}
for k, cs in acc.items():
    res[k] = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
    res[k]["dur_us_profiled"] = sum(dur[k]) / max(len(dur[k]), 1) / 1e3
print(json.dumps(res, indent=1))

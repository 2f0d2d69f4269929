#!/usr/bin/env python
"""Reduce the counter CSVs of the passes of tools/r06_profiles.sh (pmc2) to per-kernel, per-launch averages.
usage: r06_pmc_reduce.py DIR n   -> JSON on stdout"""
import collections
import csv
import glob
import json
import sys

root, n = sys.argv[1], int(sys.argv[2])
KERNELS = {"k_scale": "calibration k_scale (2^27 doubles read + written)", "k_conv_filter": "cone filter", "k_restrict<3>": "restriction 0 -> 1",
           "k_prolong_add<3>": "prolongation 1 -> 0", "k_dia_row": "level-2 block stencil (Chebyshev step)", "k_fine_tile<0>": "fine product, packed form",
           "k_fine_tile<3>": "fine Krylov product (A p)", "k_matfree_tile<2, 1>": "level-1 operator (Chebyshev step)"}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for fn in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        for k in KERNELS:
            if k in r["Kernel_Name"]:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for fn in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        for k in KERNELS:
            if k in r["Kernel_Name"]:
                dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
nd0, nd1, nd2, nel = (n + 1) ** 3, (n // 2 + 1) ** 3, (n // 4 + 1) ** 3, n ** 3
ALG = {"k_scale": 16.0 * (1 << 27), "k_conv_filter": 16.0 * nel, "k_restrict<3>": 24.0 * (nd0 + nd1), "k_prolong_add<3>": 24.0 * (2 * nd0 + nd1),
       "k_dia_row": 249.0 * 8 * nd2, "k_fine_tile<0>": 48.0 * nd0 + 8 * nel, "k_fine_tile<3>": 48.0 * nd0 + 8 * nel, "k_matfree_tile<2, 1>": 96.0 * nd1 + 64.0 * (n // 2) ** 3}
out = {"mesh": "%d^3" % n, "fetch_correction": "FETCH_SIZE x 2 (gfx950, MI355X_MICROARCH.md), KiB units; k_scale in the same passes as calibration",
       "note": "per-launch averages; durations are those of the PROFILED runs (counter collection serialises kernels)", "kernels": {}}
for k, what in KERNELS.items():
    if k not in acc:
        continue
    c = {name: sum(v) / len(v) for name, v in sorted(acc[k].items())}
    e = {"what": what, "launches": {name: len(v) for name, v in acc[k].items()}, "counters": c, "algorithmic_bytes": ALG[k],
         "dur_us_profiled": sum(dur[k]) / max(len(dur[k]), 1) / 1e3}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["hbm_bytes_per_launch"] = 1024.0 * (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"])
        e["traffic_over_algorithmic"] = e["hbm_bytes_per_launch"] / ALG[k]
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        w = c["SQ_WAVE_CYCLES"]
        e["wave_time_shares"] = {q: c[q] / w for q in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                                                       "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS") if q in c}
    if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c and c["SQ_WAVES"]:
        e["instructions_per_wave"] = {q: c[q] / c["SQ_WAVES"] for q in c if q.startswith("SQ_INSTS_")}
    out["kernels"][k] = e
print(json.dumps(out, indent=1))

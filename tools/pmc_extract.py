#!/usr/bin/env python
"""Collect FETCH_SIZE / WRITE_SIZE (KiB, per dispatch) of the two fine-level kernels from the counter CSVs of the two
separate `rocprofv3 --pmc X --kernel-trace -d <dir>/X -- python tools/pmc_traffic.py ex ey ez` passes and write the
per-launch HBM bytes (FETCH_SIZE x2 on gfx950, calibrated with k_scale; see profiles/README.md) as JSON."""
import csv
import glob
import json
import sys

root, ex, ey, ez = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
KEYS = {"spmv": ("k_fine_tile<0>", "k_fine_u4<0,"), "cheb": ("k_fine_tile<2>", "k_fine_u4<2,"), "calib": ("k_scale",)}  # second and third generation


def mean_counter(name):
    out = {k: [] for k in KEYS}
    for fn in glob.glob("%s/%s/**/*counter_collection.csv" % (root, name), recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] != name:
                continue
            for k, pat in KEYS.items():
                if any(q in r["Kernel_Name"] for q in pat):
                    out[k].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v) if v else None) for k, v in out.items()}, {k: len(v) for k, v in out.items()}


f, nf = mean_counter("FETCH_SIZE")
w, nw = mean_counter("WRITE_SIZE")
n_nd, n_el = (ex + 1) * (ey + 1) * (ez + 1), ex * ey * ez
res = {"mesh": "%dx%dx%d" % (ex, ey, ez), "launches_counted": {"fetch": nf, "write": nw},
       "calibration_kscale_2^27_doubles": {"fetch_size_kb": f["calib"], "write_size_kb": w["calib"], "true_kb_each": 8 * (1 << 27) / 1024},
       "fetch_correction": 2.0}
for k, alg in (("spmv", 48 * n_nd + 8 * n_el), ("cheb", 96 * n_nd + 8 * n_el)):
    if f[k] is None or w[k] is None:
        continue
    res[k] = {"fetch_size_kb": f[k], "write_size_kb": w[k], "hbm_bytes_per_launch": 1024.0 * (2.0 * f[k] + w[k]),
              "algorithmic_bytes": alg}
print(json.dumps(res, indent=1))

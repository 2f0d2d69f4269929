export TMPDIR=/tmp
for r in 0 1; do
rm -rf gpurun_out/pmc_remap$r
TP_XCD_REMAP=$r timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_remap$r/FETCH_SIZE -- python tools/pmc_traffic.py 128 128 128 > /dev/null 2>&1
python - <<PY
import csv, glob
acc = {}
for fn in glob.glob("gpurun_out/pmc_remap$r/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] == "FETCH_SIZE" and "k_matfree_tile" in r["Kernel_Name"]:
            acc.setdefault(r["Kernel_Name"][:30], []).append(float(r["Counter_Value"]))
dur = {}
for fn in glob.glob("gpurun_out/pmc_remap$r/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "k_matfree_tile" in r["Kernel_Name"]:
            dur.setdefault(r["Kernel_Name"][:30], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in acc: print("remap=$r", k, "fetch MB (x2 corrected)", round(2 * 1024 * sum(acc[k]) / len(acc[k]) / 1e6, 1), "us", round(sum(dur[k]) / len(dur[k]) / 1e3, 1))
PY
done

export TMPDIR=/tmp
for a in 7 8; do
echo "ablation $a"; TP_LIB=$PWD/topopt_in_petsc_amd/libtopopt_abl$a.so timeout 120 python tools/fine_ab.py 128 128 128 2>&1 | tail -n 1
TP_LIB=$PWD/topopt_in_petsc_amd/libtopopt_abl$a.so TP_TILE_KZ=32 timeout 120 python tools/fine_ab.py 256 256 256 2>&1 | tail -n 1
done
echo baseline; TP_TILE_KZ=32 timeout 120 python tools/fine_ab.py 256 256 256 2>&1 | tail -n 1
rm -rf gpurun_out/pmc_clk
timeout 240 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_clk -- python tools/pmc_traffic.py 128 128 128 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
cnt = collections.defaultdict(list); dur = collections.defaultdict(list)
for fn in glob.glob("gpurun_out/pmc_clk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        cnt[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
for fn in glob.glob("gpurun_out/pmc_clk/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        dur[r["Kernel_Name"][:40]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k in cnt:
    if "tile" in k or "k_scale" in k:
        c = sum(cnt[k]) / len(cnt[k]); d = sum(dur[k]) / len(dur[k])
        print(k, "GUI_ACTIVE", round(c), "dur_ns", round(d), "GHz", round(c / d, 3))
PY

# usage: bash tools/prof_workload.sh <workload> -- kernel-trace of the bench on another workload, prints busy time vs wall
export TMPDIR=/tmp
w=$1
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$w -- python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b_$w.json 2>/dev/null
python profiles/summarize_rocpd.py $(find gpurun_out/prof_$w -name "*.db" | head -n 1) > gpurun_out/k_$w.csv
find gpurun_out/prof_$w -name "*.db" -delete
python - <<PY
import csv, json
rows = list(csv.DictReader(open("gpurun_out/k_$w.csv")))
tot = sum(int(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
d = json.load(open("gpurun_out/b_$w.json"))
print("$w: kernels total %.1f ms over 4 steps (+setup) = %.2f ms/step; %d launches; bench wall %.2f ms/step (under profiler)" % (tot/1e6, tot/4e6, calls, d["ms_per_step"]))
for r in rows[:14]: print("  %-70s %6s calls %9.1f us avg %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY

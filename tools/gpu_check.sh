# full GPU check: parity tests, smoke, default bench (no profiling)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
grep -E "passed|failed|rc=" gpurun_out/pytest.log | tail -n 3
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_now.json"))
r = d["roofline"]
print("ms/step %.2f its %s launches %.0f | cheb %.1f us frac %.3f | spmv %.1f us frac %.3f" % (d["ms_per_step"], d["config"]["cg_its"], d["config"]["kernel_launches_per_step"], 1e3 * r["avg_launch_ms"], r["frac"], 1e3 * r["spmv"]["avg_launch_ms"], r["spmv"]["frac"]))
PY

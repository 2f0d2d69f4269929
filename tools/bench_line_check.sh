export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_bench_line.py tests/test_gpu_parity.py -q -m gpu -k "bench or vcycle or golden" 2>&1 | tail -3
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
timeout 400 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_final.json"))
r = d["roofline"]
print("ms/step %.2f its %s launches %.0f value %.4e" % (d["ms_per_step"], d["config"]["cg_its"], d["config"]["kernel_launches_per_step"], d["value"]))
print("in-step cheb %.1f us frac %.3f (%s) | back-to-back %.1f us frac %.3f | spmv %.1f us frac %.3f" % (1e3 * r["avg_launch_ms"], r["frac"], r["avg_launch_how"], 1e3 * r["back_to_back"]["avg_launch_ms"], r["back_to_back"]["frac"], 1e3 * r["spmv"]["avg_launch_ms"], r["spmv"]["frac"]))
print("256:", r["spmv256"]["spmv"]["frac"], r["spmv256"]["cheb"]["frac"])
print("cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][:160])
PY

#!/bin/bash
# the round's last word: GPU suite, smoke(), the default bench line and its exit code, then the kernel trace of the default command
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --durations=12 -p no:cacheprovider > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
tail -n 22 gpurun_out/pytest.log; tail -n 2 gpurun_out/smoke.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_default.json")); c = d["config"]; r = d["roofline"]; p = d["parity"]
print("ms %.3f value %.4g its %d launches %d  roofline %.3f (%.1f us) b2b %.3f  spmv %.1f us krylov %.1f us  l2 %.1f us" % (d["ms_per_step"], d["value"], c["cg_its"], c["kernel_launches_per_step"], r["frac"], 1e3 * r["avg_launch_ms"], r["back_to_back"]["frac"], 1e3 * r["spmv"]["avg_launch_ms"], 1e3 * r["krylov_product"]["avg_launch_ms"], 1e3 * r["level2_stencil"]["avg_launch_ms"]))
print("spmv256", {k: (round(1e3 * v["avg_launch_ms"], 1), round(v["frac"], 3)) for k, v in r["spmv256"].items() if isinstance(v, dict)})
print("parity ok", p["ok"], p["breaches"], "fx", p["fx_rel_err"], "hist", p["hist_max_rel_err_all"], "dense", p["dense_KE"]["vs_arbiter_on_KE"])
print("stated", c["stated_cycle"]["ms_per_step"], "design loop", c["design_loop"]["first"], c["design_loop"]["last"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
tail -n 3 gpurun_out/bench_default.err
bash tools/r06_profiles.sh bench cube256 2>&1 | tail -40

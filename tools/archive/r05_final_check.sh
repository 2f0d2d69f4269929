#!/bin/bash
# the round's last word: the GPU suite, smoke(), the default bench line (exit code!), all on the final code
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1300 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 | tee gpurun_out/r05_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 700 python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_line.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_line.json"))
p = d["parity"]
print("ms", d["ms_per_step"], "value", d["value"], "its", d["config"]["cg_its"], "frac", d["roofline"]["frac"], "parity ok", p["ok"], p["breaches"],
      p["arbiter"]["gpu_vs_arbiter_on_KE_eff"], p["converged"]["gpu_vs_arbiter_on_KE_eff"])
PY

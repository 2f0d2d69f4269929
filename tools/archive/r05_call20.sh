#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "arbiter_on_the_operator" 2>&1 | tail -3
timeout 700 python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_line.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_line.json"))
p = d["parity"]
print("ms", d["ms_per_step"], "value", d["value"], "its", d["config"]["cg_its"], "frac", d["roofline"]["frac"], "parity ok", p["ok"], p["breaches"])
print(json.dumps(p["arbiter"], indent=1)); print(json.dumps(p["converged"], indent=1)); print(d["cpu_baseline"]["extras"]["seconds"])
PY
timeout 300 python -m pytest tests/test_bench_line.py -x -q -m gpu -k "contract or parity" 2>&1 | tail -3

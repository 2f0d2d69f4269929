#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py::test_effective_element_matrix -x -q -m gpu 2>&1 | tail -15
TP_CG_NT=1 timeout 700 python bench.py > gpurun_out/r05_line_b.json 2> gpurun_out/r05_line_b.err; echo "bench rc=$?"; tail -3 gpurun_out/r05_line_b.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_line_b.json"))
print("ms", d["ms_per_step"], "its", d["config"]["cg_its"], "frac", d["roofline"]["frac"], "launches", d["config"]["kernel_launches_per_step"])
print(json.dumps(d.get("parity"), indent=1))
cb = d["cpu_baseline"]; print("cpu", cb["value"], cb["cores"], cb["seconds"], (cb.get("extras") or {}).get("seconds"))
PY
timeout 600 python -m pytest tests/test_bench_line.py -x -q -m gpu 2>&1 | tail -15

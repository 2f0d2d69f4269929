#!/bin/bash
# round 6, call 20: the metric mesh as 2 and 4 slabs of one GPU (host-staged gloo hooks): correctness lines with comm_time and slabs_vs_one_gpu; filter tests
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv_filter" 2>&1 | tail -3
for n in 2 4; do
  timeout 900 python bench.py --gpus $n --same-device --backend gloo --scaling strong --steps 3 --warmup 1 --no-other-scaling --budget-s 800 > gpurun_out/r06_metric_${n}slabs_same_device_line.json 2> gpurun_out/r06_metric_${n}slabs.err; echo "$n slabs rc=$?"
  python - $n <<'PY'
import json, sys
d = json.load(open("gpurun_out/r06_metric_%sslabs_same_device_line.json" % sys.argv[1])); c = d["config"]
print("128^3 as %s slabs on one GPU: ms %.2f its %d levels %s coarse %s halo_overlap %s" % (sys.argv[1], d["ms_per_step"], c["cg_its"], c["levels"], c["coarse_solve"], c["halo_overlap"]))
print("  comm per step", {k: (v["calls"], round(v["host_ms"], 2), round(v["device_ms"], 2)) for k, v in c["comm_time"]["per_step"].items()})
print("  slabs_vs_one_gpu", json.dumps(c["slabs_vs_one_gpu"]))
PY
done

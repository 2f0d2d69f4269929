#!/bin/bash
# round 6, call 7: two-rank bench tests again (timeouts raised), C3's 8-GPU geometry (16-layer slabs, 4 levels as BASELINE states) as 8 ranks on one GPU
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_line.py -x -q -m gpu -k "spawns or same_calls or distributed_run" 2>&1 | tail -5
timeout 900 python bench.py --workload c3 --nlvls 4 --cycles 1,3,1 --gpus 8 --same-device --backend gloo --scaling strong --steps 3 --warmup 1 --no-other-scaling --budget-s 800 > gpurun_out/r06_c3_8slabs_same_device_line.json 2> gpurun_out/r06_c3_8slabs_same_device.err; echo "c3 8 slabs rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_c3_8slabs_same_device_line.json")); c = d["config"]
print("c3 as 8 slabs on one GPU: ms %.2f its %d levels %s cycles %s coarse %s halo_overlap %s" % (d["ms_per_step"], c["cg_its"], c["levels"], c["cycles"], c["coarse_solve"], c["halo_overlap"]))
print("  comm_time", json.dumps(c["comm_time"])[:1200])
print("  slabs_vs_one_gpu", json.dumps(c["slabs_vs_one_gpu"]))
PY
tail -n 4 gpurun_out/r06_c3_8slabs_same_device.err

#!/bin/bash
# round 6: config 5 (512x256x256, the "10^8 DOF on 8 GPUs" case) as 8 z-slabs of ONE GPU through the gloo hooks: the
# partition, halo overlap, replicated coarse levels and the line's self-reporting (comm_time, history against one GPU)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1700 python bench.py --workload c5 --gpus 8 --same-device --backend gloo --scaling strong --steps 2 --warmup 1 --no-other-scaling --budget-s 1500 > gpurun_out/r06_c5_8slabs_same_device_line.json 2> gpurun_out/r06_c5_8slabs_same_device.err; echo "c5 8 slabs rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_c5_8slabs_same_device_line.json")); c = d["config"]
print("ms", d["ms_per_step"], "its", c["cg_its"], "levels", c["levels"], "parallelism", c["parallelism"])
print("comm_time", json.dumps(c.get("comm_time"))[:600])
print("slabs_vs_one_gpu", json.dumps(c.get("slabs_vs_one_gpu"))[:600])
PY
tail -n 4 gpurun_out/r06_c5_8slabs_same_device.err

#!/bin/bash
# round 6, call 6: the default bench trace again (reducers fixed), the slab path: two-rank bench tests, C3's 8-GPU geometry as 8 ranks on one GPU
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/r06_profiles.sh bench 2>&1 | tail -45
timeout 600 python -m pytest tests/test_bench_line.py -x -q -m gpu -k "spawns or same_calls or distributed_run" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_multirank.py -x -q -m gpu 2>&1 | tail -5
# configs[2] (256x128x128 over 8 GPUs = 16-layer slabs) at its full x-y size, as 8 ranks sharing this GPU (host-staged gloo hooks)
timeout 900 python bench.py --workload c3 --gpus 8 --same-device --backend gloo --scaling strong --steps 3 --warmup 1 --no-other-scaling --budget-s 800 > gpurun_out/r06_c3_8slabs_same_device_line.json 2> gpurun_out/r06_c3_8slabs_same_device.err; echo "c3 8 slabs rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_c3_8slabs_same_device_line.json")); c = d["config"]
print("c3 as 8 slabs on one GPU: ms %.2f its %d levels %s cycles %s coarse %s halo_overlap %s" % (d["ms_per_step"], c["cg_its"], c["levels"], c["cycles"], c["coarse_solve"], c["halo_overlap"]))
print("  comm_time", json.dumps(c["comm_time"])[:900])
print("  slabs_vs_one_gpu", json.dumps(c["slabs_vs_one_gpu"]))
PY
tail -n 4 gpurun_out/r06_c3_8slabs_same_device.err

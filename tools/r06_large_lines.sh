#!/bin/bash
# round 6: the two largest meshes at full size with the CPU worker on (256^3 with the whole parity object: the 80-bit arbiter peaks
# at 216 GB; C5 on 7 levels against the double-precision oracle alone).  The CPU worker's address space is capped below the
# container's limit (TP_CPU_WORKER_MEM_GB).
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TP_CPU_WORKER_MEM_GB=285 timeout 3000 python bench.py --workload cube256 --cpu-budget 2800 --no-cube256 --design-loop 0 > gpurun_out/r06_cube256_line.json 2> gpurun_out/r06_cube256_line.err; echo "cube256 rc=$?"
TP_CPU_WORKER_MEM_GB=280 timeout 3000 python bench.py --workload c5_deep --cpu-budget 2800 --no-parity-extras --no-cube256 --design-loop 0 > gpurun_out/r06_c5_deep_line.json 2> gpurun_out/r06_c5_deep_line.err; echo "c5_deep rc=$?"
python - <<'PY'
import json
for w in ("cube256", "c5_deep"):
    try:
        d = json.load(open("gpurun_out/r06_%s_line.json" % w)); p = d["parity"]
        print(w, "ms %.2f its %d ok %s %s fx %.2e hist10 %.2e all %.2e" % (d["ms_per_step"], d["config"]["cg_its"], p["ok"], p["breaches"], p["fx_rel_err"], p["hist_max_rel_err_first10"], p["hist_max_rel_err_all"]))
        if "arbiter" in p: print("   arbiter", p["arbiter"]["gpu_vs_arbiter_on_KE"], p["arbiter"].get("oracle_vs_arbiter_on_KE"))
        print("   cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][:160])
    except Exception as e:
        print(w, "no line:", e)
PY

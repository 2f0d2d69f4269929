#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tiled_restriction" 2>&1 | tail -3
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1 ms %.3f solve %.3f frac %.4f its %d launches %d' % (d['ms_per_step'], d['config']['solve_ms_per_step'], r['frac'], d['config']['cg_its'], d['config']['kernel_launches_per_step']))"; }
for rep in 1 2; do
  TP_RESTRICT_TILED=0 timeout 200 python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 20 --warmup 3 2>/dev/null | q untiled
  TP_RESTRICT_TILED=1 timeout 200 python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 20 --warmup 3 2>/dev/null | q tiled
done
timeout 300 python bench.py --workload c4 --no-cube256 --cpu-budget 60 > gpurun_out/r05_c4_bench_line.json 2>/dev/null; echo "c4 bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_c4_bench_line.json"))
print("c4: ms", d["ms_per_step"], "its", d["config"]["cg_its"], {k: (v.get("frac") if isinstance(v, dict) else v) for k, v in d["roofline"].items() if k in ("frac", "pde_filter", "conv_filter", "spmv")})
p = d.get("parity"); print({k: p[k] for k in p if k not in ("bounds",)})
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["problem"], d["cpu_baseline"]["cg_its"])
PY

#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$1 ms %.3f solve %.3f its %d launches %d' % (d['ms_per_step'], c['solve_ms_per_step'], c['cg_its'], c['kernel_launches_per_step']))"; }
for w in c1 c2; do
for rep in 1 2; do
  TP_CG_NT=1 timeout 200 python bench.py --workload $w --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 20 --warmup 3 2>/dev/null | q "$w nt"
  TP_CG_NT=0 timeout 200 python bench.py --workload $w --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 20 --warmup 3 2>/dev/null | q "$w plain"
  TP_CG_NT=0 TP_NO_DEFER_FACTOR=1 TP_STENCIL_OVERLAP=0 timeout 200 python bench.py --workload $w --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 20 --warmup 3 2>/dev/null | q "$w plain+nodefer"
done
done

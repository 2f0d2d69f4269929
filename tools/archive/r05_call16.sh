#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1 ms %.3f solve %.3f frac %.4f l2 %.2f us its %d launches %d' % (d['ms_per_step'], d['config']['solve_ms_per_step'], r['frac'], 1e3*r['level2_stencil']['avg_launch_ms'], d['config']['cg_its'], d['config']['kernel_launches_per_step']))"; }
for rep in 1 2 3; do
  timeout 200 python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 20 --warmup 3 2>/dev/null | q now
done

#!/bin/bash
# C5 and C2 beside their BASELINE-stated depths: what the hierarchy of the metric mesh (coarsen until the exact coarse solve fits, level 2 cycled 3x) gives
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$1: ms %.2f its %d launches %d coarse %s value %.3e' % (d['ms_per_step'], c['cg_its'], c['kernel_launches_per_step'], c['coarse_solve'], d['value']))"; }
timeout 400 python bench.py --workload c5 --no-cpu-baseline --no-cube256 --steps 2 --warmup 1 --nlvls 7 --cycles 1,3,1,1,1,1 2>/dev/null | q "c5 7 levels 1,3,1,1,1,1"
timeout 400 python bench.py --workload c5 --no-cpu-baseline --no-cube256 --steps 2 --warmup 1 --nlvls 6 --cycles 1,3,1,1,1 2>/dev/null | q "c5 6 levels 1,3,1,1,1"
timeout 200 python bench.py --workload c2 --no-cpu-baseline --no-cube256 --steps 5 --warmup 2 --nlvls 4 --cycles 1,3,1 2>/dev/null | q "c2 4 levels 1,3,1"
timeout 200 python bench.py --workload c2 --no-cpu-baseline --no-cube256 --steps 5 --warmup 2 --nlvls 5 --cycles 1,3,1,1 2>/dev/null | q "c2 5 levels 1,3,1,1"
timeout 200 python bench.py --workload c3 --no-cpu-baseline --no-cube256 --steps 3 --warmup 1 --nlvls 6 --cycles 1,2,1,1,1 2>/dev/null | q "c3 6 levels 1,2,1,1,1"

#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$1: ms %.2f solve %.2f its %d launches %d coarse %s' % (d['ms_per_step'], c['solve_ms_per_step'], c['cg_its'], c['kernel_launches_per_step'], c['coarse_solve']))"; }
B="python bench.py --workload cube256 --no-cpu-baseline --no-cube256 --steps 3 --warmup 1"
timeout 300 $B 2>/dev/null | q "cube256 default (6 levels V)"
timeout 300 $B --nlvls 6 --cycles 1,3,1,1,1 2>/dev/null | q "cube256 6 levels 1,3,1,1,1"
timeout 300 $B --nlvls 6 --cycles 1,2,1,1,1 2>/dev/null | q "cube256 6 levels 1,2,1,1,1"
B="python bench.py --workload c2_rmin008 --no-cpu-baseline --no-cube256 --steps 5 --warmup 2"
timeout 200 $B --nlvls 4 --cycles 1,3,1 2>/dev/null | q "c2 mesh 4 levels 1,3,1 (BASELINE says 3 levels)"
B="python bench.py --workload c4 --no-cpu-baseline --no-cube256 --steps 5 --warmup 2"
timeout 200 $B --nlvls 5 --cycles 1,2,2,1 2>/dev/null | q "c4 5 levels 1,2,2,1"
timeout 200 $B --nlvls 5 --cycles 1,4,1,1 2>/dev/null | q "c4 5 levels 1,4,1,1"
B="python bench.py --workload c1 --no-cpu-baseline --no-cube256 --steps 10 --warmup 2"
timeout 200 $B --nlvls 3 --cycles 1,2 2>/dev/null | q "c1 3 levels 1,2"
timeout 200 $B --nlvls 3 --cycles 1,1 2>/dev/null | q "c1 3 levels V"
timeout 200 $B --nlvls 3 --cycles 1,4 2>/dev/null | q "c1 3 levels 1,4"

#!/bin/bash
# config 4 (MBB, Helmholtz filter): depth / cycle pattern / coarse solve scan -- ms per design iteration, CG iterations
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$1: ms %.2f solve %.2f its %d launches %d coarse %s' % (d['ms_per_step'], c['solve_ms_per_step'], c['cg_its'], c['kernel_launches_per_step'], c['coarse_solve']))"; }
B="python bench.py --workload c4 --no-cpu-baseline --no-cube256 --steps 5 --warmup 2"
timeout 200 $B 2>/dev/null | q "default (6 levels V cheb45)"
timeout 200 $B --nlvls 5 --cycles 1,3,1,1 2>/dev/null | q "5 levels 1,3,1,1 direct"
timeout 200 $B --nlvls 5 --cycles 1,2,1,1 2>/dev/null | q "5 levels 1,2,1,1 direct"
timeout 200 $B --nlvls 5 --cycles 1,1,1,1 2>/dev/null | q "5 levels V direct"
timeout 200 $B --nlvls 6 --cycles 1,3,1,1,1 2>/dev/null | q "6 levels 1,3,1,1,1"
timeout 200 $B --nlvls 4 --cycles 1,3,1 2>/dev/null | q "4 levels 1,3,1 direct"
for w in c1; do
B="python bench.py --workload $w --no-cpu-baseline --no-cube256 --steps 10 --warmup 2"
timeout 200 $B 2>/dev/null | q "$w default"
timeout 200 $B --nlvls 4 --cycles 1,3,1 2>/dev/null | q "$w 4 levels 1,3,1"
timeout 200 $B --nlvls 3 --cycles 1,3 2>/dev/null | q "$w 3 levels 1,3 direct"
done

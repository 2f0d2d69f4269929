#!/bin/bash
# the headline mesh once more under the round-5 code: neighbours of the cycle 1,3,1,1
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$1: ms %.3f its %d launches %d' % (d['ms_per_step'], c['cg_its'], c['kernel_launches_per_step']))"; }
B="python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 10 --warmup 2"
timeout 200 $B 2>/dev/null | q "1,3,1,1 (default)"
for cy in 1,2,1,1 1,4,1,1 1,3,2,1 1,2,2,1 2,2,1,1; do
  timeout 200 $B --nlvls 5 --cycles $cy 2>/dev/null | q "$cy"
done
timeout 200 $B --nlvls 5 --cycles 1,3,1,1 --nsmooth 3 2>/dev/null | q "1,3,1,1 nsmooth 3"

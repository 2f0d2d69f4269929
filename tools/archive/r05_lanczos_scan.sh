#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$1: ms %.3f its %d rel %.2e' % (d['ms_per_step'], c['cg_its'], c['rel_residual']))"; }
B="python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 10 --warmup 2"
for n in 10 8 6 5 4 3; do timeout 200 $B --nlanczos $n 2>/dev/null | q "nlanczos $n"; done

#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1 ms %.3f solve %.3f frac %.4f its %d' % (d['ms_per_step'], d['config']['solve_ms_per_step'], r['frac'], d['config']['cg_its']))"; }
for rep in 1 2 3; do
  TP_CD_TRI_NT=0 timeout 200 python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 20 --warmup 3 2>/dev/null | q tri_plain
  TP_CD_TRI_NT=1 timeout 200 python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 20 --warmup 3 2>/dev/null | q tri_nt
done

#!/bin/bash
# round 6: the cycle on REAL iterates (bench.py --design-loop 60 at 128^3): does the synthetic field's optimum hold at a modulus contrast of 1e9?
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); dl=d['config']['design_loop']; its=dl['cg_its_by_iteration']; ms=dl['ms_by_iteration']
w=lambda a,b: (sum(ms[a-1:b])/(b-a+1), sum(its[a-1:b])/(b-a+1))
print('%-44s synthetic %.2f ms / %d its | real: 1-8 %.1f ms %.1f its | 9-20 %.1f ms %.1f its | 21-40 %.1f ms %.1f its | 50-60 %.1f ms %.1f its | total 60: %.0f ms' % (('$1',d['ms_per_step'],d['config']['cg_its'])+w(1,8)+w(9,20)+w(21,40)+w(50,60)+(sum(ms),)))"; }
B="python bench.py --no-cpu-baseline --no-stated-cycle --no-cube256 --steps 5 --warmup 2 --design-loop 60"
timeout 300 $B 2>/dev/null | q "default 5 lv, cheb(2), 1,3,1,1"
timeout 300 $B --nsmooth 3 2>/dev/null | q "cheb(3)"
timeout 300 $B --nsmooth 4 2>/dev/null | q "cheb(4)"
timeout 300 $B --cycles 2,3,1,1 2>/dev/null | q "cycles 2,3,1,1"
timeout 300 $B --cycles 1,2,1,1 2>/dev/null | q "cycles 1,2,1,1"
timeout 300 $B --cycles 1,3,2,1 2>/dev/null | q "cycles 1,3,2,1"
timeout 300 $B --cycles 1,1,1,1 2>/dev/null | q "V"
timeout 300 $B --cheb-hi 1.2 2>/dev/null | q "cheb-hi 1.2"
timeout 300 $B --cheb-lo 0.05 2>/dev/null | q "cheb-lo 0.05"
timeout 300 $B --nsmooth 3 --cycles 1,2,1,1 2>/dev/null | q "cheb(3), 1,2,1,1"
timeout 300 $B --fine-eig 1 2>/dev/null | q "fine level: Lanczos window"

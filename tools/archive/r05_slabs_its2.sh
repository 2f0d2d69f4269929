#!/bin/bash
# the same with a sixth level (coarsest level small enough for the exact solve again under weak scaling)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for n in 1 2 4 8; do
  timeout 600 python bench.py --gpus $n --same-device --backend gloo --steps 1 --warmup 1 --budget-s 500 --nlvls 6 --cycles 1,3,1,1,1 --no-other-scaling --no-cpu-baseline --no-cube256 --no-stated-cycle 2>gpurun_out/r05_slabs6_$n.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('N=$n weak, 6 levels 1,3,1,1,1: its %s coarse %s launches %s ms %.2f' % (c.get('cg_its'), c.get('coarse_solve'), c.get('kernel_launches_per_step'), d['ms_per_step']))" || tail -5 gpurun_out/r05_slabs6_$n.err
done
for n in 2 8; do
  timeout 600 python bench.py --gpus $n --same-device --backend gloo --steps 1 --warmup 1 --budget-s 500 --nlvls 6 --cycles 1,3,1,1,1 --no-other-scaling --coarse cheb --ncoarse 30 2>gpurun_out/r05_slabs6c_$n.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('N=$n weak, 6 levels, coarse cheb(30): its %s coarse %s launches %s' % (c.get('cg_its'), c.get('coarse_solve'), c.get('kernel_launches_per_step')))" || tail -5 gpurun_out/r05_slabs6c_$n.err
done

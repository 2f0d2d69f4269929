#!/bin/bash
# Iteration counts and coarse-solve kinds of the bench's two readings of the metric on N = 2, 4, 8 slabs -- all ranks on ONE GPU
# (gloo, host-staged halos: the times mean nothing, the iteration counts are those of a real N-GPU run) -> DESIGN 5's prediction
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for n in 2 4 8; do
  timeout 600 python bench.py --gpus $n --same-device --backend gloo --steps 1 --warmup 1 --budget-s 500 2>gpurun_out/r05_slabs_$n.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; o=d.get('other_scaling') or {}
print('N=$n weak: its %s coarse %s launches %s | strong: its %s coarse %s | comm %s' % (c.get('cg_its'), c.get('coarse_solve'), c.get('kernel_launches_per_step'), o.get('cg_its'), o.get('coarse_solve'), c.get('comm_report')))" || tail -5 gpurun_out/r05_slabs_$n.err
done

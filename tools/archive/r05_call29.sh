#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for n in 2 8; do
  timeout 600 python bench.py --gpus $n --same-device --backend gloo --steps 1 --warmup 1 --budget-s 500 2>gpurun_out/r05_w_$n.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; o=d.get('other_scaling') or {}
print('N=$n weak: levels %s cycles %s its %s coarse %s launches %s | strong: levels %s its %s coarse %s' % (c.get('levels'), c.get('cycles'), c.get('cg_its'), c.get('coarse_solve'), c.get('kernel_launches_per_step'), o.get('levels'), o.get('cg_its'), o.get('coarse_solve')))" || tail -5 gpurun_out/r05_w_$n.err
done
timeout 600 python -m pytest tests/test_bench_line.py -x -q -m gpu 2>&1 | tail -3

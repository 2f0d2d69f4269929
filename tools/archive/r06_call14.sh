#!/bin/bash
# round 6, call 14: node stencil on every larger level: tests, workloads
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fine_generations.py -x -q -m gpu -k "stencil" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_multirank.py -x -q -m gpu -k "galerkin_levels or vcycle or solve_residual or mesh_shape or config_full_size or c2_full or slab or ranks" 2>&1 | tail -4
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$1: ms %.3f its %d launches %d' % (d['ms_per_step'], c['cg_its'], c['kernel_launches_per_step']))"; }
B="--no-cpu-baseline --no-stated-cycle --no-cube256 --design-loop 0"
for w in c1 c2 c3 c4 cube256 c5_deep c2_deep; do
  timeout 400 python bench.py --workload $w --steps 5 --warmup 2 $B 2>/dev/null | q "$w node"
done
for w in cube256 c5_deep; do
  TP_DIA_NODE=0 timeout 400 python bench.py --workload $w --steps 5 --warmup 2 $B 2>/dev/null | q "$w row "
done
timeout 300 python bench.py --steps 20 --warmup 3 $B 2>/dev/null | q "128^3"

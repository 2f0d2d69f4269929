#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "galerkin_levels or solve_residual or bench_cycle or w_cycle or fine_level_lanczos or pde_filter or coarsest" 2>&1 | tail -4
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1 ms %.3f solve %.3f setup+rest %.3f its %d launches %d' % (d['ms_per_step'], d['config']['solve_ms_per_step'], d['ms_per_step']-d['config']['solve_ms_per_step'], d['config']['cg_its'], d['config']['kernel_launches_per_step']))"; }
for rep in 1 2; do
  TP_LANCZOS_FUSED=0 timeout 200 python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 20 --warmup 3 2>/dev/null | q unfused
  TP_LANCZOS_FUSED=1 timeout 200 python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 20 --warmup 3 2>/dev/null | q fused
done
TP_LANCZOS_FUSED=1 timeout 200 python bench.py --workload c4 --no-cpu-baseline --no-cube256 --steps 5 --warmup 2 2>/dev/null | q c4_fused
TP_LANCZOS_FUSED=0 timeout 200 python bench.py --workload c4 --no-cpu-baseline --no-cube256 --steps 5 --warmup 2 2>/dev/null | q c4_unfused

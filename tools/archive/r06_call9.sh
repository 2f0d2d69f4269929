#!/bin/bash
# round 6, call 9: node-per-thread stencil kernel + wide-load restriction: bitwise tests, A/B timings, iteration timeline
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fine_generations.py tests/test_gpu_parity.py -x -q -m gpu -k "stencil_kernel_per_node or galerkin_levels or vcycle or solve_residual or chebyshev or mesh_shape or bench_cycle_param or w_cycle" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_multirank.py -x -q -m gpu 2>&1 | tail -3
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('$1: ms %.3f its %d launches %d  level2 %.2f us' % (d['ms_per_step'], c['cg_its'], c['kernel_launches_per_step'], 1e3*r['level2_stencil']['avg_launch_ms']))"; }
B="python bench.py --no-cpu-baseline --no-stated-cycle --no-cube256 --steps 20 --warmup 3 --design-loop 0"
for rep in 1 2 3; do
  timeout 300 $B 2>/dev/null | q "node + wide"
  TP_DIA_NODE=0 timeout 300 $B 2>/dev/null | q "row  + wide"
  TP_LIB=$PWD/topopt_in_petsc_amd/libtopopt_amd_rn.so timeout 300 $B 2>/dev/null | q "node + narrow"
done
bash tools/r06_profiles.sh bench 2>&1 | grep -A30 "one design iteration under" | head -34
head -70 gpurun_out/r06_iteration_timeline.txt

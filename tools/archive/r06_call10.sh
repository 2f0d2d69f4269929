#!/bin/bash
# round 6, call 10: branch-free prolongation: bitwise check against the old transfer kernels, timing
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python tools/r06_transfer_bits.py topopt_in_petsc_amd/libtopopt_amd_old.so 2>&1 | tail -3
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('$1: ms %.3f its %d launches %d' % (d['ms_per_step'], c['cg_its'], c['kernel_launches_per_step']))"; }
B="python bench.py --no-cpu-baseline --no-stated-cycle --no-cube256 --steps 20 --warmup 3 --design-loop 0"
for rep in 1 2 3; do
  timeout 300 $B 2>/dev/null | q "new transfers"
  TP_LIB=$PWD/topopt_in_petsc_amd/libtopopt_amd_old.so timeout 300 $B 2>/dev/null | q "old transfers"
done
bash tools/r06_profiles.sh bench 2>&1 | grep "k_prolong_add\|k_restrict"
grep "k_prolong_add\|k_restrict" gpurun_out/r06_iteration_timeline.txt | head -20

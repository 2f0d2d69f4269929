#!/bin/bash
# every BASELINE configuration through bench.py on one GPU, final code (3 timed steps each, no CPU baseline): profiles/r06_workloads.txt
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('%-22s ms %9.3f  its %3d  launches %5d  levels %d  cycles %-10s coarse %s' % ('$1', d['ms_per_step'], c['cg_its'], c['kernel_launches_per_step'], c['levels'], c['cycles'], c['coarse_solve']))"; }
B="--no-cpu-baseline --no-stated-cycle --no-cube256 --design-loop 0 --steps 3 --warmup 2"
for w in c1 c1_stated c2 c2_deep c3 c4 c4_stated c5 c5_deep cube256 c2_rmin008 cantilever128_rmin008; do
  timeout 600 python bench.py --workload $w $B 2>/dev/null | q $w
done | tee gpurun_out/r06_workloads.txt

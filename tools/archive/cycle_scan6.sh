#!/bin/bash
# round 3, exact coarse solve: around the pattern 1,3,1,1 (level 2 cycled three times per visit of level 1, V below)
export TMPDIR=/tmp
run() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cube256 --no-stated-cycle "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('%-40s %.2f ms  its %s  launches %s  rel %.2e  %s' % (' '.join(sys.argv[1:]), d['ms_per_step'], c['cg_its'], c['kernel_launches_per_step'], c['rel_residual'], c['coarse_solve']))" "$@"; }
run --cycles 1,3,1,1
run --cycles 1,4,1,1
run --cycles 1,3,2,1
run --cycles 1,3,1,2
run --cycles 1,5,1,1
run --cycles 1,3,1,1 --rtol 1e-6
run --cycles 1,2,2,1 --rtol 1e-6
run --cycles 1,3,1,1 --workload c3 --nlvls 6

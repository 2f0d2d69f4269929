#!/bin/bash
# round 3: six-level cycles that reach the one-workgroup coarse run (coarsest 5^3 nodes) with a V into the last level
export TMPDIR=/tmp
run() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cube256 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('%-46s %.2f ms  its %s  launches %s  rel %.2e' % (' '.join(sys.argv[1:]), d['ms_per_step'], c['cg_its'], c['kernel_launches_per_step'], c['rel_residual']))" "$@"; }
run
run --nlvls 6 --cycles 1,2,2,1,1 --ncoarse 20
run --nlvls 6 --cycles 1,2,2,1,1 --ncoarse 40
run --nlvls 6 --cycles 1,2,2,2,1 --ncoarse 20
run --nlvls 6 --cycles 1,2,1,2,1 --ncoarse 30
run --nlvls 6 --cycles 1,2,2,1,2 --ncoarse 20
run --nlvls 6 --cycles 1,1,2,2,1 --ncoarse 30
run --nlvls 5 --cycles 1,2,2,1 --ncoarse 12
run --nlvls 5 --cycles 1,2,1,1 --ncoarse 30

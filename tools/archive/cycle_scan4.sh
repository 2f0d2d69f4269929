#!/bin/bash
# round 3, after the one-XCD coarse run (a coarsest-level step costs 2.3 us instead of a launch): is another cycle cheaper now?
export TMPDIR=/tmp
run() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cube256 --no-stated-cycle "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('%-46s %.2f ms  its %s  launches %s  rel %.2e' % (' '.join(sys.argv[1:]), d['ms_per_step'], c['cg_its'], c['kernel_launches_per_step'], c['rel_residual']))" "$@"; }
run
run --ncoarse 30
run --ncoarse 40
run --cycles 1,2,1,1 --ncoarse 40
run --cycles 1,1,2,1 --ncoarse 40
run --cycles 1,1,1,1 --ncoarse 40
run --cycles 1,1,1,1 --ncoarse 80
run --cycles 1,1,1,2 --ncoarse 30
run --cycles 1,2,1,2 --ncoarse 30
run --cycles 1,1,2,2 --ncoarse 30
run --cycles 1,2,2,2 --ncoarse 20

#!/bin/bash
# round 6: is the set-up bound by the Lanczos chains or by the factorisation?  step time against the number of Lanczos steps,
# with and without the deferred join of the factorisation
export TMPDIR=/tmp
for nl in 10 6 3; do
for t in "A=1" "TP_NO_DEFER_FACTOR=1"; do
  env $t timeout 400 python bench.py --nlanczos $nl --no-cube256 --no-stated-cycle --design-loop 0 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('nlanczos $nl $t ms', round(d['ms_per_step'],3), 'solve', round(c['solve_ms_per_step'],3), 'its', c['cg_its'])"
done
done

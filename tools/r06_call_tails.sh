#!/bin/bash
# round 6: A/B of the late switches (three alternating runs each): TP_LANCZOS_TAILS, TP_LANCZOS_ON_MAIN (set-up chains), TP_PROLONG_FLAT (prolongation)
export TMPDIR=/tmp
mkdir -p gpurun_out
true
true
for rep in 1 2 3; do
for t in ${SWITCHES:-A=1 TP_PROLONG_FLAT=0}; do
  env $t timeout 400 python bench.py --no-cube256 --no-stated-cycle --design-loop 0 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$t ms', round(d['ms_per_step'],3), 'solve', round(c['solve_ms_per_step'],3), 'its', c['cg_its'], 'launches', c['kernel_launches_per_step'])"
done
done

#!/bin/bash
# round 6: A/B of the late set-up switches (three alternating runs each); SWITCHES="A=1 TP_X=0 ..." overrides the list
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ -n "$TESTS" ]; then
  timeout 1200 python -m pytest $TESTS -x -q -m gpu > gpurun_out/r06_tails_tests.log 2>&1
  grep -n "passed\|failed" gpurun_out/r06_tails_tests.log
fi
for rep in 1 2 3; do
for t in ${SWITCHES:-A=1 TP_LANCZOS_TAILS=0 TP_LANCZOS_ON_MAIN=0 TP_CD_SPLIT_ENQUEUE=0}; do
  env $t timeout 400 python bench.py --no-cube256 --no-stated-cycle --design-loop 0 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$t ms', round(d['ms_per_step'],3), 'solve', round(c['solve_ms_per_step'],3), 'its', c['cg_its'], 'launches', c['kernel_launches_per_step'])"
done
done

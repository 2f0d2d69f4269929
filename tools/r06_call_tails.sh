#!/bin/bash
# round 6: the spectra chains -- in-kernel reduction tails (TP_LANCZOS_TAILS=0: second launches), small levels on the solver's
# stream (TP_LANCZOS_ON_MAIN=0: spare stream); tests that read the estimates first
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_multirank.py -x -q -m gpu > gpurun_out/r06_tails_tests.log 2>&1
grep -n "passed\|failed" gpurun_out/r06_tails_tests.log
for rep in 1 2 3; do
for t in "A=1" "TP_LANCZOS_TAILS=0" "TP_LANCZOS_ON_MAIN=0"; do
  env $t timeout 400 python bench.py --no-cube256 --no-stated-cycle --design-loop 0 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$t ms', round(d['ms_per_step'],3), 'solve', round(c['solve_ms_per_step'],3), 'its', c['cg_its'], 'launches', c['kernel_launches_per_step'])"
done
done

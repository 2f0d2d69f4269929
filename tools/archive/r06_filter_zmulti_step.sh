export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_multirank.py tests/test_golden.py -x -q -m gpu -k "filter or golden or slab or rank" > gpurun_out/r06_filter_tests.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/r06_filter_tests.log | head
for rep in 1 2 3; do
for t in A=1 TP_FILTER_ZMULTI=0; do
  env $t timeout 400 python bench.py --no-cube256 --no-stated-cycle --design-loop 0 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$t ms', round(d['ms_per_step'],3), 'solve', round(c['solve_ms_per_step'],3), 'its', c['cg_its'])"
done
done

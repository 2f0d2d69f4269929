export TMPDIR=/tmp
for rep in 1 2; do
for t in "A=1" "TP_LANCZOS_TAILS=0 TP_LANCZOS_ON_MAIN=0"; do
  env $t timeout 400 python bench.py --workload cube256 --no-cube256 --no-stated-cycle --design-loop 0 --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$t ms', round(d['ms_per_step'],3), 'solve', round(c['solve_ms_per_step'],3), 'its', c['cg_its'])"
done
done

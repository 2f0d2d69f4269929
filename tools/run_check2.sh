export TMPDIR=/tmp
one() {
timeout 300 python bench.py --workload $1 --steps 5 --warmup 1 --no-cpu-baseline --no-cube256 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('$1 $2: %.2f ms/step, its %s, launches %s, fx %.10e' % (d['ms_per_step'], c['cg_its'], c['kernel_launches_per_step'], c['fx']))"
}
for wl in cantilever128 c1 c3; do
export TP_NO_COARSE_RUN=1; one $wl launches; unset TP_NO_COARSE_RUN
for w in 16 32 64; do TP_RUN_WGS=$w one $wl run_wgs_$w; done
done

# every BASELINE configuration through bench.py on one GPU (no CPU baseline): ms per step, iterations, residual (round 5)
for w in c1 c2 c3 c4 c5 c2_rmin008 cube256 cantilever128_rmin008; do
python bench.py --workload $w --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); c=d['config']; print('%-10s' % '$w', 'ms %.2f its %d rel %.2e n_dof %d launches %d %s value %.3e' % (d['ms_per_step'], c['cg_its'], c['rel_residual'], c['n_dof'], c['kernel_launches_per_step'], c['coarse_solve'], d['value']))"
done

# iteration count and step time of the bench workload against the Chebyshev window fractions
for lo in 0.05 0.075 0.1 0.125 0.15 0.2 0.3; do for hi in 1.1; do
python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 5 --warmup 2 --cheb-lo $lo --cheb-hi $hi 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('lo $lo hi $hi', 'ms %.3f its %d rel %.3e' % (d['ms_per_step'], d['config']['cg_its'], d['config']['rel_residual']))"
done; done
python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 5 --warmup 2 --fine-eig 1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('fine-eig 1', 'ms %.3f its %d rel %.3e' % (d['ms_per_step'], d['config']['cg_its'], d['config']['rel_residual']))"

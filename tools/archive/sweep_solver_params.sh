# step time and CG iterations against the two iteration-count parameters of the V-cycle (bench.py --ncoarse / --nsmooth)
# usage: bash tools/sweep_solver_params.sh "<workloads>" "<nsmooth values>" "<ncoarse values>"
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/sweep_params.txt
: > $out
for wl in ${1:-cantilever128 c2}; do
for ns in ${2:-2 3}; do
for nc in ${3:-30 45 60 90}; do
  timeout 300 python bench.py --workload $wl --steps 5 --warmup 1 --no-cpu-baseline --no-cube256 --ncoarse $nc --nsmooth $ns 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('$wl nsmooth $ns ncoarse $nc : %.2f ms/step, CG its %s, rel res %.2e' % (d['ms_per_step'], c.get('cg_its'), c.get('rel_residual')))" >> $out
done
done
done
cat $out

export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/sweep_params_c.txt
: > $out
run() {  # workload nlvls nsmooth ncoarse steps
timeout 300 python bench.py --workload $1 --steps $5 --warmup 1 --no-cpu-baseline --no-cube256 --ncoarse $4 --nsmooth $3 --nlvls $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('$1 nlvls $2 nsmooth $3 ncoarse $4 : %.2f ms/step, CG its %s' % (d['ms_per_step'], c.get('cg_its')))" >> $out
}
for nl in 5 6; do for nc in 16 24 30 45; do run cantilever128 $nl 2 $nc 5; done; done
run cantilever128 5 1 30 5
run cantilever128 5 3 30 5
run c1 4 2 16 5
run c1 4 2 22 5
run c3 5 2 30 5
run c3 5 2 45 5
run c4 4 2 30 5
run c4 4 2 45 5
run c5 5 2 45 2
run c5 6 2 30 2
run c2 3 2 45 5
cat $out

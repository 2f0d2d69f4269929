export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/sweep_solver_params.sh "c1 c3 c4" "2 4" "30 60" > /dev/null
cp gpurun_out/sweep_params.txt gpurun_out/sweep_params_b.txt
for nl in 5; do for nc in 30 60; do
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-cube256 --ncoarse $nc --nsmooth 2 --nlvls $nl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('cantilever128 nlvls $nl nsmooth 2 ncoarse $nc : %.2f ms/step, CG its %s' % (d['ms_per_step'], c.get('cg_its')))" >> gpurun_out/sweep_params_b.txt
done; done
timeout 300 python bench.py --workload c2 --steps 5 --warmup 1 --no-cpu-baseline --no-cube256 --ncoarse 45 --nsmooth 2 --nlvls 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('c2 nlvls 4 nsmooth 2 ncoarse 45 : %.2f ms/step, CG its %s' % (d['ms_per_step'], c.get('cg_its')))" >> gpurun_out/sweep_params_b.txt
timeout 300 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-cube256 --ncoarse 60 --nsmooth 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('c5 nsmooth 2 ncoarse 60 : %.2f ms/step, CG its %s' % (d['ms_per_step'], c.get('cg_its')))" >> gpurun_out/sweep_params_b.txt
cat gpurun_out/sweep_params_b.txt

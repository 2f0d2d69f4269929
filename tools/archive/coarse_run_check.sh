export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_configs.py -q -m gpu --tb=short -k "one_launch" > gpurun_out/run_tests.log 2>&1
echo "rc $?" >> gpurun_out/run_tests.log
tail -n 12 gpurun_out/run_tests.log
one() {
timeout 300 python bench.py --workload $1 --steps 5 --warmup 1 --no-cpu-baseline --no-cube256 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('$1 $2 $3: %.2f ms/step, its %s, launches %s, fx %.10e' % (d['ms_per_step'], c['cg_its'], c['kernel_launches_per_step'], c['fx']))"
}
export TP_NO_COARSE_RUN=1; one c1 launches ""; one cantilever128 launches "--nlvls 6 --ncoarse 45"; unset TP_NO_COARSE_RUN
one c1 single ""
one c1 single "--ncoarse 45"
one c1 single "--ncoarse 90"
one cantilever128 5lv ""
for nc in 30 45 90; do one cantilever128 single "--nlvls 6 --ncoarse $nc"; done
one cantilever128 single "--nlvls 6 --ncoarse 45 --nsmooth 3"
one c3 single "--nlvls 7 --ncoarse 45"
one c4 single "--nlvls 6 --ncoarse 45"
one c2 single "--nlvls 5 --ncoarse 45"

# coarse run in one launch: parity subset, then step times with and without it
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu --tb=short -x -k "vcycle or residual_history or bench_cycle" > gpurun_out/run_tests.log 2>&1
echo "rc $?" >> gpurun_out/run_tests.log
tail -n 5 gpurun_out/run_tests.log
one() {
timeout 300 python bench.py --workload $1 --steps 5 --warmup 1 --no-cpu-baseline --no-cube256 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('$1 $2: %.2f ms/step, its %s, launches %s, fx %.10e' % (d['ms_per_step'], c['cg_its'], c['kernel_launches_per_step'], c['fx']))"
}
for wl in cantilever128 c1 c2 c3 c4; do
export TP_NO_COARSE_RUN=1; one $wl launches; unset TP_NO_COARSE_RUN
for w in 8 16 32; do TP_RUN_WGS=$w one $wl run_wgs_$w; done
done

# the tuned V-cycle parameters: parity tests, then every bench workload
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_multirank.py -q -m gpu --tb=short -k "bench_cycle" > gpurun_out/cycle_tests.log 2>&1
echo "rc $?" >> gpurun_out/cycle_tests.log
tail -n 30 gpurun_out/cycle_tests.log
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').readline())
c=d['config']
print('default: %.2f ms/step, its %s, launches %s, value %.3e' % (d['ms_per_step'], c['cg_its'], c['kernel_launches_per_step'], d['value']))
print('roofline', {k:v for k,v in d['roofline'].items() if k in ('achieved','frac','bound','traffic')})
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('sample'))
PY
for wl in c1 c2 c3 c4 c5; do
timeout 300 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null > gpurun_out/b_$wl.json
python -c "
import json
d=json.loads(open('gpurun_out/b_$wl.json').readline()); c=d['config']
print('$wl: %.2f ms/step, its %s, launches %s' % (d['ms_per_step'], c['cg_its'], c['kernel_launches_per_step']))"
done

# W-cycles (tp_elasticity_set_cycles): parity test, then step times at 128^3 for a few cycle patterns
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 40 python -m pytest tests/test_gpu_parity.py -q -m gpu --tb=line -k "w_cycles" 2>&1 | tail -4
one() {
timeout 20 python bench.py --workload $1 --steps 3 --warmup 1 --no-cpu-baseline --no-cube256 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('$1 $2: %.2f ms/step, its %s, launches %s' % (d['ms_per_step'], c['cg_its'], c['kernel_launches_per_step']))"
}
one cantilever128 "--cycles 1,2,2,1 --ncoarse 10"
one cantilever128 "--cycles 1,2,2,1 --ncoarse 20"
one cantilever128 "--cycles 1,2,1,1"
one cantilever128 "--nlvls 6 --cycles 1,2,2,2,1 --ncoarse 20"
one c3 "--cycles 1,2,2,2,2,1 --ncoarse 20"
one c2 "--nlvls 3 --cycles 2,1 --ncoarse 20"

# iteration count and step time of the bench workload against smoothing steps / cycle patterns (exact coarse solve)
run() { python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 5 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('%-40s' % '$*', 'ms %.3f its %d rel %.3e launches %d' % (d['ms_per_step'], d['config']['cg_its'], d['config']['rel_residual'], d['config']['kernel_launches_per_step']))"; }
run
run --nsmooth 1
run --nsmooth 1 --cycles 1,4,1,1
run --nsmooth 1 --cycles 2,3,1,1
run --nsmooth 1 --cycles 1,3,2,1
run --nsmooth 3
run --nsmooth 3 --cycles 1,2,1,1
run --nsmooth 3 --cycles 1,1,1,1
run --cycles 1,4,1,1
run --cycles 1,2,1,1
run --cycles 1,2,2,1

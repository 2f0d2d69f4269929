run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('ms %.3f its %d solve %.2f fx %.12e rel %.6e launches %d' % (d['ms_per_step'], d['config']['cg_its'], d['config']['solve_ms_per_step'], d['config']['fx'], d['config']['rel_residual'], d['config']['kernel_launches_per_step']))"; }
export TP_BENCH_MEASURE_S=0.1
run A=1
run TP_CD_INVERT_COLUMNS=1
run A=1

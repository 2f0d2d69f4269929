run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('ms %.3f its %d l2 stencil %.1f us' % (d['ms_per_step'], d['config']['cg_its'], 1e3*r['level2_stencil']['avg_launch_ms']))"; }
export TP_BENCH_MEASURE_S=0.2
run A=1
run TP_NO_DIA_SYM=1
run TP_DIA_SPLIT=1
run TP_DIA_SPLIT=9
run A=1

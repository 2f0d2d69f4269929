# does the CPU baseline running first (256 OpenMP threads for ~14 s) slow the GPU phases that follow?
export TP_BENCH_MEASURE_S=0.2
p() { python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1: ms %.3f solve %.2f in-step cheb %.1f us b2b %.1f' % (d['ms_per_step'], d['config']['solve_ms_per_step'], 1e3*r['avg_launch_ms'], 1e3*r['back_to_back']['avg_launch_ms']))"; }
python bench.py --no-cpu-baseline --no-cube256 --steps 20 --warmup 5 2>/dev/null | p "gpu only   "
python bench.py --no-cube256 --steps 20 --warmup 5 2>/dev/null | p "cpu first  "
python bench.py --no-cpu-baseline --no-cube256 --steps 20 --warmup 5 2>/dev/null | p "gpu only   "
python bench.py --no-cube256 --steps 20 --warmup 5 2>/dev/null | p "cpu first  "
python bench.py --no-cube256 --steps 20 --warmup 30 2>/dev/null | p "cpu first, warm-up 30"

# the CPU baseline (oracle, 128^3, the bench's cycle) against thread count and placement on the GPU box's host
W="python bench.py --cpu-baseline-worker /tmp/cb.json 128x128x128 1e-5 0 128 128 128 6440067 400 5 2 20 1,3,1,1 1"
p() { python -c "
import json; d=json.load(open('/tmp/cb.json')); print('$1', 'seconds %.2f' % d['seconds'], {k: round(v,2) for k,v in d['phase_seconds'].items()}, 'mf %.2f' % d['matrix_free']['seconds'])"; }
$W; p "default(256)"
TP_CPU_THREADS=128 OMP_PROC_BIND=spread OMP_PLACES=cores $W; p "128 spread cores"
TP_CPU_THREADS=256 OMP_PROC_BIND=close OMP_PLACES=threads $W; p "256 close threads"
TP_CPU_THREADS=64 OMP_PROC_BIND=spread OMP_PLACES=cores $W; p "64 spread cores"

#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "solve_residual or bench_cycle or w_cycle or coarsest or pde_filter" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_multirank.py -x -q -m gpu -k "two_ranks_one_gpu or bench_w_cycle" 2>&1 | tail -3
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$1: ms %.3f its %d' % (d['ms_per_step'], c['cg_its']))"; }
B="python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 20 --warmup 3"
for rep in 1 2 3; do timeout 200 $B 2>/dev/null | q "128^3"; done
timeout 200 python bench.py --workload c4 --no-cpu-baseline --no-cube256 --steps 10 --warmup 2 2>/dev/null | q c4
timeout 200 python bench.py --workload c1 --no-cpu-baseline --no-cube256 --steps 20 --warmup 3 2>/dev/null | q c1

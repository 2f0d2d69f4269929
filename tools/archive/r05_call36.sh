#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_golden.py -x -q -m gpu -k "pde or c4 or golden or product_kf" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_multirank.py -x -q -m gpu -k "two_ranks_one_gpu or three_ranks" 2>&1 | tail -3
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('$1: ms %.2f solve %.2f its %d pde_filter %s' % (d['ms_per_step'], c['solve_ms_per_step'], c['cg_its'], {k: r['pde_filter'][k] for k in ('avg_launch_ms','frac')}))"; }
for rep in 1 2; do
TP_NO_PDE_STENCIL=1 timeout 200 python bench.py --workload c4 --no-cpu-baseline --no-cube256 --steps 10 --warmup 2 2>/dev/null | q "c4 gather form"
timeout 200 python bench.py --workload c4 --no-cpu-baseline --no-cube256 --steps 10 --warmup 2 2>/dev/null | q "c4 stencil form"
done

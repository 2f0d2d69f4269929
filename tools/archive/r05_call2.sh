#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/r05_fx_diag.py 128 2>&1 | tail -40
for d in 1 0 1 0; do
  if [ $d = 1 ]; then unset TP_NO_DEFER_FACTOR; else export TP_NO_DEFER_FACTOR=1; fi
  TP_CG_NT=1 timeout 200 python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('defer=$d ms %.3f solve_ms %.3f frac %.4f its %d' % (d['ms_per_step'], d['config']['solve_ms_per_step'], d['roofline']['frac'], d['config']['cg_its']))"
done
unset TP_NO_DEFER_FACTOR
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "coarsest or give_up or solve_residual or bench_cycle or w_cycle" 2>&1 | tail -5

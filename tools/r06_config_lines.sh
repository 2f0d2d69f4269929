#!/bin/bash
# round 6: full-size oracle parity lines of C2 / C3 / C4 (as configured and with the Helmholtz filter at rtol 1e-13): CPU baseline + parity object on
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); p=d.get('parity') or {}; c=d['config']
print(sys.argv[1], 'ms %.3f its %d ok %s breaches %s' % (d['ms_per_step'], c['cg_its'], p.get('ok'), p.get('breaches')))
print('   fx_rel_err %s hist10 %s hist_all %s gx %s' % (p.get('fx_rel_err'), p.get('hist_max_rel_err_first10'), p.get('hist_max_rel_err_all'), p.get('gx_abs_err')))
if 'arbiter' in p: print('   gpu_vs_arbiter_on_KE', p['arbiter']['gpu_vs_arbiter_on_KE'], 'dense', (p.get('dense_KE') or {}).get('vs_oracle_on_KE'))
if 'converged' in p: print('   converged gpu_vs_oracle', p['converged']['gpu_vs_oracle'])
pf=d['roofline'].get('pde_filter')
if pf: print('   pde', pf['kernel'][:40], json.dumps(pf.get('solver_comparison')))
print('   cpu', d['cpu_baseline'].get('sample'), d['cpu_baseline'].get('value'))
PY
}
for w in c2 c3; do
  timeout 1500 python bench.py --workload $w --cpu-budget 1200 --no-cube256 > gpurun_out/r06_${w}_line.json 2> gpurun_out/r06_${w}_line.err; echo "$w rc=$?"
  show gpurun_out/r06_${w}_line.json; tail -n 3 gpurun_out/r06_${w}_line.err
done
timeout 1500 python bench.py --workload c4 --cpu-budget 1200 --no-cube256 --pde-rtol 1e-13 > gpurun_out/r06_c4_tight_line.json 2> gpurun_out/r06_c4_tight_line.err; echo "c4 tight rc=$?"
show gpurun_out/r06_c4_tight_line.json; tail -n 3 gpurun_out/r06_c4_tight_line.err
timeout 1500 python bench.py --workload c4 --cpu-budget 1200 --no-cube256 > gpurun_out/r06_c4_line.json 2> gpurun_out/r06_c4_line.err; echo "c4 rc=$?"
show gpurun_out/r06_c4_line.json; tail -n 3 gpurun_out/r06_c4_line.err

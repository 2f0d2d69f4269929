#!/bin/bash
# round 6, call 3: the Krylov operator's translation column/row -- parity tests, A/B cost of the variants, C2/C3 lines again
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fine_generations.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -6
timeout 600 python -m pytest tests/test_bench_line.py tests/test_gpu_configs.py -x -q -m gpu -k "c2_full or bench_prints or exits_nonzero or design_loop" 2>&1 | tail -6
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; s=r.get('spmv256') or {}
print('$1: ms %.3f its %d  cheb in-step %.1f us  spmv128 %.1f us krylov128 %.1f us  spmv256 %s cheb256 %s krylov256 %s' % (d['ms_per_step'], c['cg_its'], 1e3*r['avg_launch_ms'], 1e3*r['spmv']['avg_launch_ms'], 1e3*r['krylov_product']['avg_launch_ms'], s.get('spmv',{}).get('avg_launch_ms'), s.get('cheb',{}).get('avg_launch_ms'), s.get('krylov_product',{}).get('avg_launch_ms')))"; }
B="python bench.py --no-cpu-baseline --no-stated-cycle --steps 20 --warmup 3 --design-loop 0"
for rep in 1 2; do
  for v in "" _nokry; do
    if [ -z "$v" ]; then timeout 300 $B 2>/dev/null | q "default"; else TP_LIB=$PWD/topopt_in_petsc_amd/libtopopt_amd$v.so timeout 300 $B 2>/dev/null | q "$v"; fi
  done
done
show() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); p=d.get('parity') or {}; c=d['config']
print(sys.argv[1], 'ms %.3f its %d ok %s breaches %s' % (d['ms_per_step'], c['cg_its'], p.get('ok'), p.get('breaches')))
print('   fx_rel_err %s hist10 %s hist_all %s gx %s' % (p.get('fx_rel_err'), p.get('hist_max_rel_err_first10'), p.get('hist_max_rel_err_all'), p.get('gx_abs_err')))
if 'arbiter' in p:
    for k in ('gpu_vs_arbiter_on_KE_eff','gpu_vs_arbiter_on_KE','oracle_vs_arbiter_on_KE','arbiter_on_KE_eff_vs_on_KE'): print('   ',k, p['arbiter'][k])
    print('    dense', (p.get('dense_KE') or {}).get('vs_oracle_on_KE'))
if 'converged' in p: print('   converged gpu_vs_oracle', p['converged']['gpu_vs_oracle'])
print('   design_loop', json.dumps(c.get('design_loop'))[:1500])
PY
}
timeout 1500 python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; echo "default rc=$?"
show gpurun_out/r06_bench_default.json; tail -n 3 gpurun_out/r06_bench_default.err
for w in c2 c3; do
  timeout 1500 python bench.py --workload $w --cpu-budget 1200 --no-cube256 > gpurun_out/r06_${w}_line.json 2> gpurun_out/r06_${w}_line.err; echo "$w rc=$?"
  show gpurun_out/r06_${w}_line.json; tail -n 3 gpurun_out/r06_${w}_line.err
done

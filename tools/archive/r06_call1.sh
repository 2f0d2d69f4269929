#!/bin/bash
# round 6, call 1: the three translation residues in the packed element matrix -- parity tests, A/B cost, the default line
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fine_generations.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -5
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('$1: ms %.3f its %d  cheb in-step %.1f us b2b %.1f us  spmv256 %s' % (d['ms_per_step'], c['cg_its'], 1e3*r['avg_launch_ms'], 1e3*r['back_to_back']['avg_launch_ms'], json.dumps(r.get('spmv256'))[:300]))"; }
B="python bench.py --no-cpu-baseline --no-stated-cycle --steps 20 --warmup 3"
for rep in 1 2; do
  timeout 300 $B 2>/dev/null | q "36 values"
  TP_LIB=$PWD/topopt_in_petsc_amd/libtopopt_amd_no36.so timeout 300 $B 2>/dev/null | q "33 values"
done
timeout 900 python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_bench_default.json'))
p=d['parity']
print('ms', d['ms_per_step'], 'its', d['config']['cg_its'])
print('ok', p['ok'], p['breaches'])
for k in ('fx_rel_err','hist_max_rel_err_first10','hist_max_rel_err_all'): print(k, p[k])
print('arbiter', json.dumps(p['arbiter'], indent=0)[:1500])
print('converged', json.dumps(p['converged'])[:1500])
print('dense', json.dumps(p.get('dense_KE')))
print('em', json.dumps(p['element_matrix']))
PY
tail -n 5 gpurun_out/r06_bench_default.err

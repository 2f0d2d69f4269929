export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('$1: ms %.3f its %d  cheb in-step %.1f us b2b %.1f' % (d['ms_per_step'], c['cg_its'], 1e3*r['avg_launch_ms'], 1e3*r['back_to_back']['avg_launch_ms']))"; }
B="python bench.py --no-cpu-baseline --no-stated-cycle --no-cube256 --steps 20 --warmup 3 --design-loop 0"
for rep in 1 2; do
timeout 300 $B 2>/dev/null | q "default"
for v in 1 2 3 4 6; do TP_LIB=$PWD/topopt_in_petsc_amd/libtopopt_amd_st$v.so timeout 300 $B 2>/dev/null | q "store aux $v"; done
done

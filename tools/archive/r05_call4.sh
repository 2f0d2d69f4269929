#!/bin/bash
# NT-hint experiments: (a) in-step variants of the CG update / restriction, (b) the 256^3 fine kernels with nt stores / nt epilogue loads
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r.get('spmv256')
print('$1 ms %.3f frac %.4f b2b %.4f' % (d['ms_per_step'], r['frac'], r['back_to_back']['frac']), ('spmv256 %.4f (%.1f us) cheb256 %.4f (%.1f us)' % (s['spmv']['frac'], 1e3*s['spmv']['avg_launch_ms'], s['cheb']['frac'], 1e3*s['cheb']['avg_launch_ms'])) if s else '')"; }
for rep in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 20 --warmup 3 2>/dev/null | q base
  TP_CG_NTS=1 timeout 200 python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 20 --warmup 3 2>/dev/null | q cg_nts
  TP_NT_RESTRICT=1 timeout 200 python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 20 --warmup 3 2>/dev/null | q nt_restrict
  TP_CG_NT=0 timeout 200 python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 20 --warmup 3 2>/dev/null | q cg_nt_off
done
for lib in "" topopt_in_petsc_amd/libtopopt_amd_nt1.so topopt_in_petsc_amd/libtopopt_amd_nt2.so; do
  TP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 300 python bench.py --no-cpu-baseline --no-stated-cycle --steps 3 --warmup 1 2>/dev/null | q "lib=${lib:-default}"
done

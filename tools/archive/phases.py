#!/usr/bin/env python
"""Where does a design iteration of the bench workload spend its time?  Wall times (device synchronised) of the
phases: assembly + Galerkin + spectra, CG iterations (by limiting max_it), one preconditioner application, filters.
usage: phases.py [workload-args of bench: --nlvls N --cycles a,b,c --ncoarse K --nsmooth S]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import topopt_in_petsc_amd as tp

p = argparse.ArgumentParser()
p.add_argument("--el", default="128,128,128")
p.add_argument("--nlvls", type=int, default=5)
p.add_argument("--nsmooth", type=int, default=2)
p.add_argument("--ncoarse", type=int, default=20)
p.add_argument("--cycles", default="1,2,2,1")
a = p.parse_args()
ex, ey, ez = [int(v) for v in a.el.split(",")]
h = 1.0 / ey
grid = tp.Grid(ex + 1, ey + 1, ez + 1, h)
le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=a.nlvls, nsmooth=a.nsmooth, ncoarse=a.ncoarse, rtol=1e-5))
if a.cycles:
    le.set_cycles([int(v) for v in a.cycles.split(",")])
le.SetUpLoadAndBC()
flt = tp.Filter(grid, 1, 2.56 * h)
x = grid.synth_density(12345)
xt, xp, df, dg = grid.elem_vec(), grid.elem_vec(), grid.elem_vec(), grid.elem_vec()
flt.FilterProject(x, xt, xp)


def wall(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


t_asm = wall(lambda: le.AssembleStiffnessMatrix(xp, 1e-9, 1.0, 3.0))


def solve(maxit):
    le.L.tp_elasticity_set_tolerances(le.handle, 1e-5, 1e-50, 1e5, maxit)
    le.U.zero_()
    le.KSPSolve()


res = {}
for m in (1, 2, 4, 8, 13, 200):
    res[m] = wall(lambda: solve(m), 3)
    its = le.last_its
    print("solve max_it %3d: %.3f ms (its %d)" % (m, res[m], its))
b = grid.node_vec(3).normal_()
t_pc = wall(lambda: le.precond(b), 10)
t_flt = wall(lambda: flt.FilterProject(x, xt, xp), 10)
t_obj = wall(lambda: le.ComputeObjectiveConstraintsSensitivities(df, dg, xp, 1e-9, 1.0, 3.0, 0.12), 2)
print("assemble+Galerkin+spectra %.3f ms | per CG iteration %.3f ms | one preconditioner application %.3f ms | filter %.3f ms | whole objective call %.3f ms"
      % (t_asm, (res[13] - res[1]) / 12.0, t_pc, t_flt, t_obj))

#!/usr/bin/env python
"""Wall time (device synchronised) of the set-up part of a design iteration of the bench workload: SIMP moduli,
Galerkin operators, spectra chains / coarse factorisation.  usage: r04_setup.py [n_repeats]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import topopt_in_petsc_amd as tp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ex = ey = ez = 128
h = 1.0 / ey
grid = tp.Grid(ex + 1, ey + 1, ez + 1, h)
le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=5, nsmooth=2, ncoarse=20, rtol=1e-5, coarse_direct=1))
le.set_cycles([1, 3, 1, 1])
le.SetUpLoadAndBC()
xp = grid.synth_density(12345)
for _ in range(3):
    le.AssembleStiffnessMatrix(xp, 1e-9, 1.0, 3.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    le.AssembleStiffnessMatrix(xp, 1e-9, 1.0, 3.0)
    torch.cuda.synchronize()
print("GPU_MAX_HW_QUEUES=%s  set-up %.3f ms per assembly" % (os.environ.get("GPU_MAX_HW_QUEUES", "default"), 1e3 * (time.perf_counter() - t0) / n))
le.U.zero_()
its = le.KSPSolve()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    le.U.zero_()
    le.KSPSolve()
torch.cuda.synchronize()
print("solve alone %.3f ms, %d its" % (1e3 * (time.perf_counter() - t0) / 5, its))

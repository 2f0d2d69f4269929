#!/usr/bin/env python
"""Wall time of the set-up part of a design iteration (assembly + Galerkin + spectra) against the Lanczos step count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import topopt_in_petsc_amd as tp
ex = ey = ez = 128
h = 1.0 / ey
grid = tp.Grid(ex + 1, ey + 1, ez + 1, h)
xp = grid.synth_density(12345)
for nl in (10, 5, 1):
    le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=5, nsmooth=2, ncoarse=20, rtol=1e-5, nlanczos=nl))
    le.set_cycles([1, 2, 2, 1])
    le.SetUpLoadAndBC()
    for _ in range(2):
        le.AssembleStiffnessMatrix(xp, 1e-9, 1.0, 3.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        le.AssembleStiffnessMatrix(xp, 1e-9, 1.0, 3.0)
    torch.cuda.synchronize()
    print("nlanczos %2d: %.3f ms per assemble (env %s)" % (nl, 1e3 * (time.perf_counter() - t0) / 6, {k: v for k, v in os.environ.items() if k.startswith("TP_")}))
    del le

#!/usr/bin/env python
"""Only the set-up part of a design iteration (assembly, Galerkin operators, spectra), for kernel traces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import topopt_in_petsc_amd as tp
ex = ey = ez = 128
h = 1.0 / ey
grid = tp.Grid(ex + 1, ey + 1, ez + 1, h)
le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=5, nsmooth=2, ncoarse=20, rtol=1e-5, coarse_direct=1))
le.set_cycles([1, 3, 1, 1])
le.SetUpLoadAndBC()
xp = grid.synth_density(12345)
for _ in range(8):
    le.AssembleStiffnessMatrix(xp, 1e-9, 1.0, 3.0)
torch.cuda.synchronize()

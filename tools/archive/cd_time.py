#!/usr/bin/env python
"""Set-up and solve time at 128^3 with the exact coarse solve on / off (SolverOptions.coarse_direct)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import topopt_in_petsc_amd as tp
ex = ey = ez = 128
grid = tp.Grid(ex + 1, ey + 1, ez + 1, 1.0 / ey)
xp = grid.synth_density(12345)
for cd in ((1,) if os.environ.get("TP_CD_STAGES") else (1, 0)):
    le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=5, nsmooth=2, ncoarse=20, rtol=1e-5, coarse_direct=cd))
    le.set_cycles([1, 2, 2, 1])
    le.SetUpLoadAndBC()
    for _ in range(2):
        le.AssembleStiffnessMatrix(xp, 1e-9, 1.0, 3.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        le.AssembleStiffnessMatrix(xp, 1e-9, 1.0, 3.0)
    torch.cuda.synchronize()
    t_as = (time.perf_counter() - t0) / 6
    le.U.zero_(); le.KSPSolve()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        le.U.zero_()
        its = le.KSPSolve()
    torch.cuda.synchronize()
    t_so = (time.perf_counter() - t0) / 4
    print("coarse_direct %d (active rows %d): set-up %.3f ms, solve %.3f ms (%d its, %.3f ms per iteration)" % (cd, le.coarse_direct_active(), 1e3 * t_as, 1e3 * t_so, its, 1e3 * t_so / its))
    del le

#!/usr/bin/env python
"""Launch time of the fine-level kernels against the problem size (fixed 128x128 cross-section, growing z):
separates the per-launch fixed cost from the per-element cost."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import topopt_in_petsc_amd as tp

ex = ey = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for ez in (4, 8, 16, 32, 64, 128):
    h = 1.0 / ey
    grid = tp.Grid(ex + 1, ey + 1, ez + 1, h)
    le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=1))
    le.SetUpLoadAndBC()
    x = grid.synth_density()
    le.AssembleStiffnessMatrix(x, 1e-9, 1.0, 3.0)
    u = grid.node_vec(3).normal_()
    y = torch.zeros_like(u)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn, reps=40):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    t_apply = timed(lambda: le.MatMult(u, y))
    t8 = timed(lambda: le.smooth(0, u, y, 8, False), 10)
    t0 = timed(lambda: le.smooth(0, u, y, 0, False), 10)
    print("ez=%4d  elements %8d  apply %7.1f us   cheb %7.1f us" % (ez, ex * ey * ez, t_apply, (t8 - t0) / 8), flush=True)
    del le, grid

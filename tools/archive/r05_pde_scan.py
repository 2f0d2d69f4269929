"""Helmholtz filter at config 4's mesh: time and iterations of one FilterProject against the depth / step counts of its multigrid."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import topopt_in_petsc_amd as tp

ex, ey, ez = 192, 64, 64
h = 1.0 / ey
grid = tp.Grid(ex + 1, ey + 1, ez + 1, h)
x = grid.synth_density()
xt, xp = grid.elem_vec(), grid.elem_vec()
for nlv, nsm, nco in ((3, 2, 10), (3, 2, 6), (3, 2, 4), (3, 2, 2), (2, 2, 10), (2, 2, 4), (3, 3, 4), (3, 1, 4), (2, 3, 4), (1, 2, 2)):
    try:
        f = tp.Filter(grid, 2, 2.56 * h, tp.SolverOptions(nlvls=nlv, rtol=1e-8, dtol=1e3, max_it=60, nsmooth=nsm, ncoarse=nco))
    except Exception as e:
        print("nlv %d: %r" % (nlv, e))
        continue
    for _ in range(3):
        f.FilterProject(x, xt, xp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        f.FilterProject(x, xt, xp)
    torch.cuda.synchronize()
    print("levels %d nsmooth %d ncoarse %2d: %.3f ms per FilterProject, %d iterations, rel res %.2e" % (
        nlv, nsm, nco, 1e2 * (time.perf_counter() - t0), f.last_pde_solve()[0], f.last_pde_solve()[1]), flush=True)
    f.close()

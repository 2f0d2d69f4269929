"""Is a 6-step Lanczos estimate (instead of 10) safe for the Chebyshev windows?  CG iterations of the bench cycle at 128^3 on
three density fields -- the bench's synthetic one, the uniform start of an optimisation, a nearly binary one (contrast 1e9) --
with 10, 8, 6, 5, 4 steps."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import topopt_in_petsc_amd as tp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ex = ey = ez = n
h = 1.0 / ey
grid = tp.Grid(ex + 1, ey + 1, ez + 1, h)
flt = tp.Filter(grid, 1, 2.56 * h)
x0 = grid.synth_density(12345)
fields = {"synthetic": x0, "uniform 0.12": torch.full_like(x0, 0.12),
          "nearly binary": (x0 > 0.25).double() * 0.999 + 0.001, "binary, filtered": None}
xt, xp = grid.elem_vec(), grid.elem_vec()
flt.FilterProject(fields["nearly binary"], xt, xp)
fields["binary, filtered"] = xp.clone()
for name, x in fields.items():
    row = []
    for nl in (10, 8, 6, 5, 4):
        le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=5, rtol=1e-5, nsmooth=2, ncoarse=20, coarse_direct=1, nlanczos=nl, max_it=100))
        le.set_cycles([1, 3, 1, 1])
        le.SetUpLoadAndBC()
        try:
            its = le.SolveState(x, 1e-9, 1.0, 3.0)
            row.append("%d: %d its (lam1 %.4f)" % (nl, its, le.level_lambda(1)))
        except Exception as e:
            row.append("%d: %s" % (nl, type(e).__name__))
        le.close()
    print("%-18s %s" % (name, " | ".join(row)), flush=True)

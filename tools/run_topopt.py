#!/usr/bin/env python
"""End-to-end run of the reference's optimisation loop on the MI355X path.
usage: run_topopt.py ex ey ez nlvls n_iter [filter [nsmooth ncoarse]]   (e.g. 128 128 128 5 20 1 2 45: the cycle of bench.py)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import topopt_in_petsc_amd as tp

ex, ey, ez, nlv, nit = [int(v) for v in sys.argv[1:6]]
flt = int(sys.argv[6]) if len(sys.argv) > 6 else 1
h = 1.0 / ey
opt = tp.TopOpt(nxyz=(ex + 1, ey + 1, ez + 1), xc=(0, ex * h, 0, 1, 0, ez * h), nlvls=nlv, rmin=2.56 * h, filter=flt,
                solver=tp.SolverOptions(nlvls=nlv, **(dict(nsmooth=int(sys.argv[7]), ncoarse=int(sys.argv[8])) if len(sys.argv) > 8 else {})))
print("# %dx%dx%d elements, %d DOF, %d MG levels, filter %d, rmin %.4f" % (ex, ey, ez, 3 * (ex + 1) * (ey + 1) * (ez + 1), nlv, flt, 2.56 * h))
for it in range(nit):
    r = opt.step(verbose=True)
    print("State solver:  iter: %i, rerr.: %e | MMA inner its: %d" % (r["ksp_its"], r["ksp_rerr"], r["mma_inner"]), flush=True)

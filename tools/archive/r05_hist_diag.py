"""What is the 5.9e-11 left between the GPU's residual history and the arbiter's on KE_eff at 128^3?  Hypothesis: the coarse
hierarchy -- the library builds its Galerkin operators from KE (its level-1 operator from KE's own packed form), the arbiter's
hierarchy was built from KE_eff.  Test: the arbiter with the FINE operator from KE_eff and the hierarchy from KE."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import topopt_in_petsc_amd as tp
from oracle import arbiter as arb
from oracle import oracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ex = ey = ez = n
nlv = 5 if n >= 128 else 4
cyc = [1, 3, 1, 1][: nlv - 1]
nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
grid = tp.Grid(nx, ny, nz, h)
le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=nlv, rtol=1e-5, nsmooth=2, ncoarse=20, coarse_direct=1))
le.set_cycles(cyc)
le.SetUpLoadAndBC()
flt = tp.Filter(grid, 1, 2.56 * h)
x = grid.synth_density(12345)
xt, xp, df, dg = grid.elem_vec(), grid.elem_vec(), grid.elem_vec(), grid.elem_vec()
flt.FilterProject(x, xt, xp)
fx_g, _ = le.ComputeObjectiveConstraintsSensitivities(df, dg, xp, 1e-9, 1.0, 3.0, 0.12, hist_cap=64)
hg = np.array(le.last_hist)
KE, kf = le.KE, le.KE_effective()
xpn = xp.cpu().numpy()
N, R = orc.cantilever_bc(nx, ny, nz, h)
E = orc.simp(xpn)
mg = arb.MG(nx, ny, nz, 3, nlv, 2, 20)
mg.set_coarse_direct(True)
mg.set_cycles(cyc)
err = lambda hist: float(np.abs(hg / np.asarray(hist, dtype=np.float64)[: len(hg)] - 1).max())
for tag, K_all, K_fine in (("hierarchy and fine operator from KE_eff", kf, None), ("hierarchy from KE, fine operator from KE_eff", KE, kf),
                           ("hierarchy and fine operator from KE", KE, None)):
    mg.assemble(K_all, E, N)
    if K_fine is not None:
        mg.reassemble_fine(K_fine)
    U, its, hist = mg.solve(arb.f64(R * N), rtol=1e-5)
    fx = arb.compliance_sens(nx, ny, nz, KE, U, xpn)[0]
    print("arbiter, %-46s its %d/%d  GPU hist vs it: max %.3e  first 3: %s   fx %.3e" % (
        tag, its, le.last_its, err(hist), ["%.1e" % abs(hg[i] / float(hist[i]) - 1) for i in range(3)], abs(fx_g / float(fx) - 1)), flush=True)

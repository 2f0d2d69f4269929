"""Where does the 1.6e-10 between the GPU's and the oracle's CONVERGED compliance at 128^3 come from?
Compares, on the bench's own case: x, the filtered density, the moduli, b, the converged U, and the objective evaluated
crosswise (GPU objective on the oracle's U, oracle objective on the GPU's U)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import topopt_in_petsc_amd as tp
from oracle import oracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ex = ey = ez = n
nlv = 5 if n >= 128 else 4
cyc = [1, 3, 1, 1][: nlv - 1]
nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
grid = tp.Grid(nx, ny, nz, h)
le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=nlv, rtol=1e-12, nsmooth=2, ncoarse=20, coarse_direct=1))
le.set_cycles(cyc)
le.SetUpLoadAndBC()
flt = tp.Filter(grid, 1, 2.56 * h)
x = grid.synth_density(12345)
xt, xp, df, dg = grid.elem_vec(), grid.elem_vec(), grid.elem_vec(), grid.elem_vec()
flt.FilterProject(x, xt, xp)
fx_g, gx_g = le.ComputeObjectiveConstraintsSensitivities(df, dg, xp, 1e-9, 1.0, 3.0, 0.12, hist_cap=64)
Ug = le.U.cpu().numpy()
print("gpu: its %d fx %.17g" % (le.last_its, fx_g))

xo = orc.synth_density(ex, ey, ez, h)
of = orc.Filter(nx, ny, nz, h, 2.56 * h)
xto, xpo = of.project(1, xo)
print("x      gpu vs cpu: max abs %.3e" % np.abs(x.cpu().numpy() - xo).max())
d = xp.cpu().numpy() - xpo
print("xPhys  gpu vs cpu: max abs %.3e, max rel %.3e, mean rel (signed) %.3e" % (np.abs(d).max(), np.abs(d / xpo).max(), (d / xpo).mean()))
hs_g, hs_c = flt.Hs().cpu().numpy(), of.hs()
print("Hs     gpu vs cpu: max rel %.3e" % np.abs(hs_g / hs_c - 1).max())
KE = orc.hex8_ke_box(h, h, h, 0.3)
print("KE     gpu vs cpu: max abs %.3e" % np.abs(le.KE - KE).max())
N, R = orc.cantilever_bc(nx, ny, nz, h)
print("N, RHS gpu vs cpu: %.3e %.3e" % (np.abs(le.N.cpu().numpy() - N).max(), np.abs(le.RHS.cpu().numpy() - R).max()))
t0 = time.time()
mg = orc.MG(nx, ny, nz, 3, nlv, 2, 20)
mg.set_coarse_direct(True)
mg.set_cycles(cyc)
# (1) the oracle on ITS OWN filtered density, (2) the oracle on the GPU's filtered density
for tag, xpc in (("cpu xPhys", xpo), ("gpu xPhys", xp.cpu().numpy())):
    mg.assemble(KE, orc.simp(xpc), N)
    U, its, hist = mg.solve(R * N, rtol=1e-12)
    fo, go, dfo, dgo = orc.compliance_sens(nx, ny, nz, KE, U, xpc)
    print("oracle on %s: its %d fx %.17g   gpu/oracle - 1 = %.3e   (%.0f s)" % (tag, its, fo, fx_g / fo - 1, time.time() - t0))
    print("   U gpu vs oracle: max rel-to-max %.3e;  dfdx %.3e" % (rel(Ug, U), rel(df.cpu().numpy(), dfo)))
    # crosswise objective
    f_cg = orc.compliance_sens(nx, ny, nz, KE, Ug, xpc)[0]
    le.U.copy_(torch.from_numpy(U).cuda())
    f_gc = le.Objective(torch.from_numpy(np.ascontiguousarray(xpc)).cuda(), 1e-9, 1.0, 3.0, 0.12)[0]
    le.U.copy_(torch.from_numpy(Ug).cuda())
    print("   oracle objective on GPU U: %.17g (vs oracle %.3e);  GPU objective on oracle U: %.17g (vs oracle %.3e)" % (f_cg, f_cg / fo - 1, f_gc, f_gc / fo - 1))
    # true residuals with the oracle's operator
    b = R * N
    for nm, V in (("U_gpu", Ug), ("U_oracle", U)):
        r = b - mg.apply(0, V)
        print("   ||b - K_oracle %s|| / ||b|| = %.3e   b.U = %.17g" % (nm, np.linalg.norm(r) / np.linalg.norm(b), float(b @ V)))
    # the GPU operator against the oracle's on the oracle's U
    yg = le.MatMult(torch.from_numpy(U).cuda()).cpu().numpy() if tag == "gpu xPhys" else None
    if yg is not None:
        yo = mg.apply(0, U)
        print("   K_gpu U vs K_oracle U: max rel-to-max %.3e" % rel(yg, yo))

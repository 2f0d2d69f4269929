#!/usr/bin/env python
"""Freezes outputs of the CPU oracle as regression fixtures (SURVEY.md 8(c) "golden fixtures to
commit").  These are ORACLE outputs, not reference outputs: they guard the oracle against drift and
give the GPU tests a second, file-based target.  Run from the repo root:
    python tests/golden/make_oracle_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402


def case(ex, ey, ez, nlv, rtol):
    nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
    KE = orc.hex8_ke_box(h, h, h, 0.3)
    N, R = orc.cantilever_bc(nx, ny, nz, h)
    x = orc.synth_density(ex, ey, ez, h)
    flt = orc.Filter(nx, ny, nz, h, 2.56 * h)
    xt, xp = flt.project(1, x)
    mg = orc.MG(nx, ny, nz, 3, nlv)
    mg.assemble(KE, orc.simp(xp), N)
    rng = np.random.default_rng(42)
    v = rng.standard_normal(3 * nx * ny * nz)
    U, its, hist = mg.solve(R * N, rtol=rtol, maxit=300)
    fx, gx, df, dg = orc.compliance_sens(nx, ny, nz, KE, U, xp)
    dff = flt.gradient(1, x, xt, df)
    pf = orc.PDEFilter(nx, ny, nz, h, 2.56 * h, nlv=min(nlv, 3), nsmooth=2, ncoarse=10)
    xpde, its_p, _ = pf.apply(x)
    return dict(dims=np.array([ex, ey, ez, nlv]), rtol=rtol, x=x, xTilde=xt, Hs=flt.hs(), v=v, Kv=mg.apply(0, v),
                lam=np.array([mg.lam(l) for l in range(nlv)]), lam_min=mg.lam_min(nlv - 1), its=its, hist=hist,
                U=U.astype(np.float32), fx=fx, gx=gx, dfdx=df, dfdx_filtered=dff, xpde=xpde, its_pde=its_p)


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden")
    np.savez_compressed(os.path.join(out, "oracle_16x8x8.npz"), **case(16, 8, 8, 3, 1e-8))
    c1 = case(48, 24, 24, 4, 1e-5)   # BASELINE config C1: iteration-1-like scalars only
    np.savez_compressed(os.path.join(out, "oracle_c1_scalars.npz"), dims=c1["dims"], its=c1["its"], hist=c1["hist"],
                        fx=c1["fx"], gx=c1["gx"], lam=c1["lam"], lam_min=c1["lam_min"], its_pde=c1["its_pde"])
    print("written", os.listdir(out))

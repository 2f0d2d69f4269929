"""CPU experiment (oracle level matrices, scipy products): what would an EXACT coarsest-level solve buy against the 20-step
Chebyshev run?  CG iterations for several cycle shapes.   usage: cycle_experiment_exact_coarse.py ex ey ez nlv"""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, scipy.sparse.linalg as spla
from oracle import oracle as orc
ex, ey, ez, nlv = [int(v) for v in sys.argv[1:5]]
nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
x = orc.synth_density(ex, ey, ez, h)
flt = orc.Filter(nx, ny, nz, h, 2.56 * h)
xt, xp = flt.project(1, x)
KE = orc.hex8_ke_box(h, h, h, 0.3)
N, R = orc.cantilever_bc(nx, ny, nz, h)
mg = orc.MG(nx, ny, nz, 3, nlv, 2, 20)
mg.assemble(KE, orc.simp(xp), N)
A = [mg.csr(l) for l in range(nlv)]
dinv = [1.0 / a.diagonal() for a in A]
lam = [mg.lam(l) for l in range(nlv)]
lmin = mg.lam_min(nlv - 1)
b = R * N
lu = spla.splu(A[-1].tocsc())
print("levels:", [a.shape[0] for a in A], "coarsest window", lmin, 1.1 * lam[-1], flush=True)


def cheb(l, lo, hi, rhs, x0, k, zero):
    theta, delta = 0.5 * (hi + lo), 0.5 * (hi - lo)
    sigma = theta / delta; rho = 1.0 / sigma
    r = rhs.copy() if zero else rhs - A[l] @ x0
    d = dinv[l] * r / theta; x = x0 + d
    for _ in range(1, k):
        rn = 1.0 / (2 * sigma - rho)
        r = rhs - A[l] @ x
        d = rn * rho * d + 2 * rn / delta * (dinv[l] * r)
        x = x + d; rho = rn
    return x


def its(gam, ncoarse, ns=2):
    visits = [0]
    def cyc(l, rhs):
        if l == nlv - 1:
            visits[0] += 1
            return lu.solve(rhs) if ncoarse == 0 else cheb(l, lmin, 1.1 * lam[l], rhs, np.zeros_like(rhs), ncoarse, True)
        xl = cheb(l, 0.1 * lam[l], 1.1 * lam[l], rhs, np.zeros_like(rhs), ns, True)
        for g in range(gam[l] if l + 1 < nlv - 1 else 1):
            xl = xl + mg.prolong(l, cyc(l + 1, mg.restrict(l, rhs - A[l] @ xl)))
        return cheb(l, 0.1 * lam[l], 1.1 * lam[l], rhs, xl, ns, False)
    xk = np.zeros_like(b); r = b.copy(); bn = np.linalg.norm(b); n = 0
    z = cyc(0, r); p = z.copy(); rz = r @ z
    while n < 200:
        w = A[0] @ p; a = rz / (p @ w); xk += a * p; r -= a * w; n += 1
        if np.linalg.norm(r) <= 1e-5 * bn: break
        z = cyc(0, r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return n, np.linalg.norm(r) / bn, visits[0] / (n)


for gam in ([1, 2, 2, 1][:nlv - 1], [1, 2, 1, 1][:nlv - 1], [1, 1, 2, 1][:nlv - 1], [1, 1, 1, 1][:nlv - 1]):
    for nc in (20, 40, 0):
        n, rel, v = its(gam, nc)
        print("cycles %s coarse %s: its %d rel %.2e coarse visits per V %.1f" % (gam, "exact" if nc == 0 else "cheb %d" % nc, n, rel, v), flush=True)

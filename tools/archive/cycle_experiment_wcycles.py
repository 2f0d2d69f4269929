"""CPU experiment: CG iterations with level-dependent Chebyshev step counts (point Jacobi), oracle level matrices."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, scipy.sparse.linalg as spla
from oracle import oracle as orc
ex, ey, ez, nlv, nc = [int(v) for v in sys.argv[1:6]]
nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
x = orc.synth_density(ex, ey, ez, h)
flt = orc.Filter(nx, ny, nz, h, 2.56 * h)
xt, xp = flt.project(1, x)
KE = orc.hex8_ke_box(h, h, h, 0.3)
N, R = orc.cantilever_bc(nx, ny, nz, h)
mg = orc.MG(nx, ny, nz, 3, nlv, 2, nc)
mg.assemble(KE, orc.simp(xp), N)
A = [mg.csr(l) for l in range(nlv)]
dinv = [1.0 / a.diagonal() for a in A]
lam = [mg.lam(l) for l in range(nlv)]
lam[0] = mg.lam(0)
lmin = mg.lam_min(nlv - 1)
b = R * N

def cheb(l, lo, hi, rhs, x0, k, zero):
    if k == 0: return x0.copy()
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

def its_for(ns, ncoarse, gamma=1, exact=False):
    lu = spla.splu(A[-1].tocsc()) if exact else None
    def cyc(l, rhs):
        if l == nlv - 1:
            return lu.solve(rhs) if exact else cheb(l, lmin, 1.1 * lam[l], rhs, np.zeros_like(rhs), ncoarse, True)
        xl = cheb(l, 0.1 * lam[l], 1.1 * lam[l], rhs, np.zeros_like(rhs), ns[l], True)
        for g in range(gamma if l > 0 else 1):
            xl = xl + mg.prolong(l, cyc(l + 1, mg.restrict(l, rhs - A[l] @ xl)))
        return cheb(l, 0.1 * lam[l], 1.1 * lam[l], rhs, xl, ns[l], False)
    xk = np.zeros_like(b); r = b.copy(); bn = np.linalg.norm(b); its = 0
    z = cyc(0, r); p = z.copy(); rz = r @ z
    while its < 200:
        w = A[0] @ p; a = rz / (p @ w); xk += a * p; r -= a * w; its += 1
        if np.linalg.norm(r) <= 1e-5 * bn: break
        z = cyc(0, r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return its
import itertools
def its_g(ns, ncoarse, gam):
    def cyc(l, rhs):
        if l == nlv - 1:
            return cheb(l, lmin, 1.1 * lam[l], rhs, np.zeros_like(rhs), ncoarse, True)
        xl = cheb(l, 0.1 * lam[l], 1.1 * lam[l], rhs, np.zeros_like(rhs), ns[l], True)
        for g in range(gam[l]):
            xl = xl + mg.prolong(l, cyc(l + 1, mg.restrict(l, rhs - A[l] @ xl)))
        return cheb(l, 0.1 * lam[l], 1.1 * lam[l], rhs, xl, ns[l], False)
    xk = np.zeros_like(b); r = b.copy(); bn = np.linalg.norm(b); its = 0
    z = cyc(0, r); p = z.copy(); rz = r @ z
    while its < 200:
        w = A[0] @ p; a = rz / (p @ w); xk += a * p; r -= a * w; its += 1
        if np.linalg.norm(r) <= 1e-5 * bn: break
        z = cyc(0, r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return its
ns = [2] * (nlv - 1)
for gam, ncs in (([1,2,2,2], (45, 20, 10, 5)), ([1,1,2,2], (45, 10)), ([1,1,1,2], (45, 10)), ([1,2,1,1], (45,)), ([1,2,2,1], (45, 10)), ([2,1,1,1], (45,)), ([1,1,1,1], (45,))):
    gam = (gam + [gam[-1]] * nlv)[:nlv - 1]
    for ncx in ncs:
        print("gamma per level", gam, "nc", ncx, ": its", its_g(ns, ncx, gam), flush=True)

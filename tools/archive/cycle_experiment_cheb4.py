"""CPU experiment (oracle level matrices, scipy products): CG iterations of the bench cycle with the smoothing polynomial
taken as first-kind Chebyshev on [lo, 1.1] lambda (the reference's PETSc default, lo = 0.1) or as the fourth-kind
Chebyshev polynomial of Lottes (2023) on (0, 1.1 lambda] -- the same three-term form, other coefficients.
usage: cycle_experiment_cheb4.py ex ey ez nlv ncoarse  cycles(comma)"""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from oracle import oracle as orc
ex, ey, ez, nlv, nc = [int(v) for v in sys.argv[1:6]]
gam = [int(v) for v in sys.argv[6].split(",")]
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
lmin = mg.lam_min(nlv - 1)
b = R * N


def coefs(kind, lo, hi, k):
    """(c1, c2) per step: d = c1 d + c2 dinv r, x += d"""
    out = []
    if kind == 1:
        theta, delta = 0.5 * (hi + lo), 0.5 * (hi - lo)
        sigma = theta / delta
        rho = 1.0 / sigma
        out.append((0.0, 1.0 / theta))
        for _ in range(1, k):
            rn = 1.0 / (2 * sigma - rho)
            out.append((rn * rho, 2 * rn / delta))
            rho = rn
    else:
        out.append((0.0, 4.0 / (3.0 * hi)))
        for i in range(1, k):
            out.append(((2.0 * i - 1.0) / (2.0 * i + 3.0), (8.0 * i + 4.0) / ((2.0 * i + 3.0) * hi)))
    return out


def cheb(l, cf, rhs, x0, zero):
    x = x0.copy()
    d = np.zeros_like(rhs)
    for i, (c1, c2) in enumerate(cf):
        r = rhs.copy() if (zero and i == 0) else rhs - A[l] @ x
        d = c1 * d + c2 * (dinv[l] * r)
        x = x + d
    return x


def its(kind, lo_f, ns):
    cf = [coefs(kind, lo_f * lam[l], 1.1 * lam[l], ns) for l in range(nlv)]
    cc = coefs(1, lmin, 1.1 * lam[-1], nc)
    def cyc(l, rhs):
        if l == nlv - 1:
            return cheb(l, cc, rhs, np.zeros_like(rhs), True)
        xl = cheb(l, cf[l], rhs, np.zeros_like(rhs), True)
        for g in range(gam[l]):
            xl = xl + mg.prolong(l, cyc(l + 1, mg.restrict(l, rhs - A[l] @ xl)))
        return cheb(l, cf[l], rhs, xl, False)
    xk = np.zeros_like(b); r = b.copy(); bn = np.linalg.norm(b); n = 0
    z = cyc(0, r); p = z.copy(); rz = r @ z
    while n < 200:
        w = A[0] @ p; a = rz / (p @ w); xk += a * p; r -= a * w; n += 1
        if np.linalg.norm(r) <= 1e-5 * bn: break
        z = cyc(0, r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return n, np.linalg.norm(r) / bn


for kind, lo, ns in ((1, 0.1, 2), (4, 0.0, 2), (1, 0.2, 2), (1, 0.3, 2), (1, 0.1, 3), (4, 0.0, 3), (1, 0.1, 1), (4, 0.0, 1)):
    n, rel = its(kind, lo, ns)
    print("kind %d lo %.1f nsmooth %d: its %d rel %.2e" % (kind, lo, ns, n, rel), flush=True)

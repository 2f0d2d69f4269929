"""CPU experiment (oracle level matrices, scipy products, exact coarsest-level solve): CG iterations of the bench cycle with
per-level smoothing step counts and cycle counts.   usage: cycle_experiment_levelsteps.py ex ey ez nlv"""
import sys, time
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
mg.set_coarse_direct(True)
mg.assemble(KE, orc.simp(xp), N)
A = [mg.csr(l) for l in range(nlv)]
dinv = [1.0 / a.diagonal() for a in A]
lam = [mg.lam(l) for l in range(nlv - 1)] + [1.0]
b = R * N
lu = spla.splu(A[-1].tocsc())
print("levels:", [a.shape[0] for a in A], flush=True)


def cheb(l, rhs, x0, k, zero):
    if k == 0:
        return x0.copy()
    lo, hi = 0.1 * lam[l], 1.1 * lam[l]
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


def its(gam, ns):
    apps = [0] * nlv
    def cyc(l, rhs, x0, zero):
        if l == nlv - 1:
            apps[l] += 1
            return lu.solve(rhs)
        xl = cheb(l, rhs, x0, ns[l], zero)
        apps[l] += ns[l] - (1 if zero else 0) + 1
        rc = mg.restrict(l, rhs - A[l] @ xl)
        xc = None
        for g in range(gam[l] if l + 1 < nlv - 1 else 1):   # PCMGMCycle_Private: same right-hand side, from the iterate
            xc = cyc(l + 1, rc, np.zeros_like(rc) if g == 0 else xc, g == 0)
        xl = xl + mg.prolong(l, xc)
        apps[l] += ns[l]
        return cheb(l, rhs, xl, ns[l], False)
    xk = np.zeros_like(b); r = b.copy(); bn = np.linalg.norm(b); n = 0
    z = cyc(0, r, np.zeros_like(r), True); p = z.copy(); rz = r @ z
    while n < 100:
        w = A[0] @ p; a = rz / (p @ w); xk += a * p; r -= a * w; n += 1
        if np.linalg.norm(r) <= 1e-5 * bn: break
        z = cyc(0, r, np.zeros_like(r), True); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return n, [a / n for a in apps]


cost = [60.0, 36.0, 12.4, 5.0, 13.0]   # us per operator application of a level at 128^3 (coarsest: per solve)
variants = [([1, 3, 1, 1], [2, 2, 2, 2]), ([1, 3, 1, 1], [2, 2, 1, 2]), ([1, 3, 1, 1], [2, 2, 1, 1]), ([1, 4, 1, 1], [2, 2, 1, 2]),
            ([1, 3, 1, 1], [2, 1, 2, 2]), ([1, 3, 1, 1], [2, 2, 3, 2]), ([1, 3, 1, 1], [2, 3, 1, 2]), ([1, 2, 2, 1], [2, 2, 2, 2]),
            ([1, 3, 2, 1], [2, 2, 1, 1]), ([1, 5, 1, 1], [2, 2, 1, 1])]
for gam, ns in variants:
    t = time.time()
    n, apps = its(gam[:nlv - 1], ns[:nlv - 1] + [0])
    est = sum(c * a for c, a in zip(cost, apps)) * n
    print("cycles %s steps %s: its %d, applications per iteration %s, kernel-time estimate %.2f ms (%.0f s)" % (gam[:nlv - 1], ns[:nlv - 1], n, ["%.1f" % a for a in apps], est / 1e3, time.time() - t), flush=True)

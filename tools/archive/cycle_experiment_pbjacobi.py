"""CPU experiment: CG iterations of the V-cycle with point Jacobi vs 3x3 point-block Jacobi inside the Chebyshev smoothers
(PETSc: -mg_levels_pc_type jacobi | pbjacobi), on the oracle's level matrices."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
from oracle import oracle as orc
ex, ey, ez, nlv, ns, nc = [int(v) for v in sys.argv[1:7]]
nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
x = orc.synth_density(ex, ey, ez, h)
flt = orc.Filter(nx, ny, nz, h, 2.56 * h)
xt, xp = flt.project(1, x)
KE = orc.hex8_ke_box(h, h, h, 0.3)
N, R = orc.cantilever_bc(nx, ny, nz, h)
mg = orc.MG(nx, ny, nz, 3, nlv, ns, nc)
mg.assemble(KE, orc.simp(xp), N)
A = [mg.csr(l) for l in range(nlv)]
b = R * N

def block_inv(Al):
    n = Al.shape[0] // 3
    D = np.zeros((n, 3, 3))
    Ac = Al.tocsr()
    for r in range(3):
        for c in range(3):
            D[:, r, c] = np.asarray(Ac[r::3, :][:, c::3].diagonal()).ravel()
    Di = np.linalg.inv(D)
    return lambda v: np.einsum('nij,nj->ni', Di, v.reshape(n, 3)).ravel()

def point_inv(Al):
    d = 1.0 / Al.diagonal()
    return lambda v: d * v

def extremes(Al, M):
    n = Al.shape[0]
    op = spla.LinearOperator((n, n), matvec=lambda v: M(Al @ v))
    lmax = spla.eigs(op, k=1, which='LM', tol=1e-6, maxiter=5000)[0][0].real
    return lmax
def lmin_of(Al, M):
    n = Al.shape[0]
    op = spla.LinearOperator((n, n), matvec=lambda v: M(Al @ v))
    return spla.eigs(op, k=1, which='SM', tol=1e-6, maxiter=20000)[0][0].real if n > 60 else np.linalg.eigvals(np.array([M(Al @ e) for e in np.eye(n)]).T).real.min()

def cheb(Al, M, lo, hi, rhs, x0, k, zero):
    theta, delta = 0.5 * (hi + lo), 0.5 * (hi - lo)
    sigma = theta / delta
    rho = 1.0 / sigma
    x = x0.copy()
    r = rhs.copy() if zero else rhs - Al @ x
    d = M(r) / theta
    x = x + d
    for _ in range(1, k):
        rn = 1.0 / (2 * sigma - rho)
        r = rhs - Al @ x
        d = rn * rho * d + 2 * rn / delta * M(r)
        x = x + d
        rho = rn
    return x

def run(kind):
    M = [(block_inv if kind == 'pb' else point_inv)(Al) for Al in A]
    lam = [extremes(A[l], M[l]) for l in range(nlv)]
    lmin = lmin_of(A[-1], M[-1])
    def cyc(l, rhs):
        if l == nlv - 1:
            return cheb(A[l], M[l], lmin, 1.1 * lam[l], rhs, np.zeros_like(rhs), nc, True)
        xl = cheb(A[l], M[l], 0.1 * lam[l], 1.1 * lam[l], rhs, np.zeros_like(rhs), ns, True)
        xl = xl + mg.prolong(l, cyc(l + 1, mg.restrict(l, rhs - A[l] @ xl)))
        return cheb(A[l], M[l], 0.1 * lam[l], 1.1 * lam[l], rhs, xl, ns, False)
    xk = np.zeros_like(b); r = b.copy(); bn = np.linalg.norm(b); its = 0
    z = cyc(0, r); p = z.copy(); rz = r @ z
    while np.linalg.norm(r) > 1e-5 * bn and its < 200:
        w = A[0] @ p; a = rz / (p @ w); xk += a * p; r -= a * w; its += 1
        if np.linalg.norm(r) <= 1e-5 * bn: break
        z = cyc(0, r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return its, [round(v, 3) for v in lam], lmin
for kind in ('pt', 'pb'):
    t0 = time.time(); print(kind, run(kind), "%.0f s" % (time.time() - t0), flush=True)
print("oracle its", mg.solve(b, rtol=1e-5)[1])

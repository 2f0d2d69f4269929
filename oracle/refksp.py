"""refksp.py -- CPU restatement of the solver configuration the reference HARD-CODES, on the oracle's assembled level
matrices (numpy / scipy; small meshes).  TEST INFRASTRUCTURE: only tests/ may import it; the product path
(csrc/refksp.h, tp_solver_opts.ksp_mode = 1) never does.

What is restated, and from where:
  * the configuration: LinearElasticity.cc:620-746 (FGMRES(100), rtol 1e-5, non-zero guess; PCMG V-cycle, Galerkin;
    level smoothers GMRES(4) x 4 iterations + PCSOR; coarse GMRES(30) <= 30 iterations, rtol 1e-8, PCSOR) and
    PDEFilter.cc:276-378 (FGMRES(20), rtol 1e-8, dtol 1e3, <= 60; smoothers GMRES(1) x 1 + PCJACOBI; coarse GMRES(10)
    <= 10, PCJACOBI);
  * the algorithms are PETSc 3.11's (un-vendored third party, makefile_ref:1), restated from its published
    descriptions -- PARITY UNPINNED against PETSc itself, like the rest of the solver oracle:
      KSPGMRES   left-preconditioned, classical Gram-Schmidt without refinement, Givens rotations, convergence on the
                 preconditioned residual norm relative to the initial one (zero guess);
      KSPFGMRES  right-preconditioned (flexible), the same orthogonalisation, convergence on the recurrence residual
                 norm relative to ||b|| when the initial guess is non-zero (KSPConvergedDefault);
      PCMG       multiplicative V-cycle; PCMGSetLevels installs KSPConvergedSkip on the level smoothers (they run
                 their max_it), the coarse solver keeps the default test;
      PCSOR      MatSOR with the defaults: one local symmetric sweep, omega = 1, zero initial guess, i.e.
                 z = (D + U)^-1 D (D + L)^-1 r in the natural (DMDA) row order.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def ssor_apply(A, r):
    """one symmetric Gauss-Seidel sweep from a zero guess: forward (D + L) y = r, then backward (D + U) z = r - L y"""
    A = A.tocsr()
    DL = sp.tril(A, 0, format="csr")
    DU = sp.triu(A, 0, format="csr")
    L = sp.tril(A, -1, format="csr")
    y = spla.spsolve_triangular(DL, r, lower=True)
    return spla.spsolve_triangular(DU, r - L @ y, lower=False)


def gauss_seidel_sweep_loops(A, b, x, backward=False):
    """the same sweep row by row (pure Python, tiny meshes): x_i <- (b_i - sum_{j != i} a_ij x_j) / a_ii in place"""
    A = A.tocsr()
    n = A.shape[0]
    x = x.copy()
    rows = range(n - 1, -1, -1) if backward else range(n)
    for i in rows:
        lo, hi = A.indptr[i], A.indptr[i + 1]
        cols, vals = A.indices[lo:hi], A.data[lo:hi]
        d = vals[cols == i][0]
        x[i] = x[i] + (b[i] - vals @ x[cols]) / d
    return x


class _Hess:
    """the Hessenberg least-squares problem of GMRES kept triangular with Givens rotations"""

    def __init__(self, m, beta):
        self.R = np.zeros((m + 1, m))
        self.cs, self.sn = np.zeros(m), np.zeros(m)
        self.g = np.zeros(m + 1)
        self.g[0] = beta

    def column(self, j, h):
        h = h.copy()
        for i in range(j):
            a = self.cs[i] * h[i] + self.sn[i] * h[i + 1]
            h[i + 1] = -self.sn[i] * h[i] + self.cs[i] * h[i + 1]
            h[i] = a
        d = np.hypot(h[j], h[j + 1])
        self.cs[j], self.sn[j] = (h[j] / d, h[j + 1] / d) if d > 0 else (1.0, 0.0)
        h[j] = d
        self.g[j + 1] = -self.sn[j] * self.g[j]
        self.g[j] = self.cs[j] * self.g[j]
        self.R[: j + 1, j] = h[: j + 1]
        return abs(self.g[j + 1])

    def solve(self, k):
        return np.linalg.solve(self.R[:k, :k], self.g[:k]) if k else np.zeros(0)


def gmres_left(A, pc, b, x0, m, maxit, rtol=0.0, atol=0.0, dtol=np.inf, test=False):
    """KSPGMRES(m) with the left preconditioner pc(r); returns (x, its, preconditioned residual norms)"""
    x = np.zeros_like(b) if x0 is None else x0.copy()
    its, hist, done = 0, [], maxit < 1
    ttol = rnorm0 = 0.0
    while not done:
        r = pc(b if (x0 is None and its == 0) else b - A @ x)
        beta = np.linalg.norm(r)
        if its == 0:
            rnorm0, ttol = beta, max(rtol * beta, atol)
            hist.append(beta)
        if beta == 0.0 or (test and beta <= ttol) or its >= maxit:
            break
        V = np.zeros((m + 1, b.size))
        V[0] = r / beta
        H = _Hess(m, beta)
        j = 0
        while j < m and its < maxit:
            w = pc(A @ V[j])
            h = np.zeros(m + 2)
            h[: j + 1] = V[: j + 1] @ w          # classical Gram-Schmidt: all dot products with the unmodified w
            w = w - h[: j + 1] @ V[: j + 1]
            hn = np.linalg.norm(w)
            h[j + 1] = hn
            if hn > 0:
                V[j + 1] = w / hn
            res = H.column(j, h)
            its += 1
            j += 1
            hist.append(res)
            if test and (res <= ttol or res >= dtol * rnorm0):
                done = True
            if hn == 0.0:
                done = True
            if done:
                break
        if its >= maxit:
            done = True
        x = x + H.solve(j) @ V[:j]
    return x, its, np.array(hist)


class RefSolver:
    """FGMRES around the PCMG V-cycle with GMRES level solvers; `mg` is an assembled oracle.MG (level matrices, Q1
    transfer).  pc: 1 = PCSOR, 0 = PCJACOBI."""

    def __init__(self, mg, restart=100, rtol=1e-5, atol=1e-50, dtol=1e5, max_it=200, nsmooth=4, ncoarse=30, smooth_pc=1,
                 coarse_pc=1, coarse_restart=30, coarse_rtol=1e-8):
        self.mg, self.nlv = mg, mg.nlv
        self.A = [mg.csr(l) for l in range(mg.nlv)]
        self.dinv = [1.0 / A.diagonal() for A in self.A]
        self.o = dict(restart=restart, rtol=rtol, atol=atol, dtol=dtol, max_it=max_it, nsmooth=nsmooth, ncoarse=ncoarse,
                      smooth_pc=smooth_pc, coarse_pc=coarse_pc, coarse_restart=coarse_restart, coarse_rtol=coarse_rtol)
        self.coarse_its = 0

    def pc(self, l, kind):
        return (lambda r: ssor_apply(self.A[l], r)) if kind == 1 else (lambda r: self.dinv[l] * r)

    def smooth(self, l, b, x0):
        o = self.o
        return gmres_left(self.A[l], self.pc(l, o["smooth_pc"]), b, x0, o["nsmooth"], o["nsmooth"])[0]

    def vcycle(self, l, b):
        o = self.o
        if l == self.nlv - 1:
            x, its, _ = gmres_left(self.A[l], self.pc(l, o["coarse_pc"]), b, None, o["coarse_restart"], o["ncoarse"],
                                   o["coarse_rtol"], o["atol"], o["dtol"], test=True)
            self.coarse_its += its
            return x
        x = self.smooth(l, b, None)
        rc = self.mg.restrict(l, b - self.A[l] @ x)
        x = x + self.mg.prolong(l, self.vcycle(l + 1, rc))
        return self.smooth(l, b, x)

    def solve(self, b, x0=None):
        """returns (x, its, recurrence residual norms ||b - A x_k||, k = 0..its)"""
        o, A = self.o, self.A[0]
        x = np.zeros_like(b) if x0 is None else x0.copy()
        m = max(1, o["restart"])
        bnorm = np.linalg.norm(b)
        its, hist, done = 0, [], False
        ttol = ref = 0.0
        while not done:
            r = b - A @ x
            beta = np.linalg.norm(r)
            if its == 0:
                ref = bnorm if bnorm > 0 else beta
                ttol = max(o["rtol"] * ref, o["atol"])
                hist.append(beta)
            if beta <= ttol or its >= o["max_it"]:
                break
            V, Z = [r / beta], []
            H = _Hess(m, beta)
            j = 0
            while j < m and its < o["max_it"]:
                Z.append(self.vcycle(0, V[j]))
                w = A @ Z[j]
                Vm = np.array(V)
                h = np.zeros(m + 2)
                h[: j + 1] = Vm @ w
                w = w - h[: j + 1] @ Vm
                hn = np.linalg.norm(w)
                h[j + 1] = hn
                V.append(w / hn if hn > 0 else w)
                res = H.column(j, h)
                its += 1
                j += 1
                hist.append(res)
                if res <= ttol or not res <= o["dtol"] * ref or hn == 0.0:
                    done = True
                    break
            if its >= o["max_it"]:
                done = True
            x = x + H.solve(j) @ np.array(Z[:j])
        return x, its, np.array(hist)

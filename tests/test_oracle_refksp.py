"""CPU pins of oracle/refksp.py, the restatement of the solver configuration the reference hard-codes (FGMRES + PCMG with
GMRES / SOR level solvers, LinearElasticity.cc:620-746; GMRES / Jacobi in PDEFilter.cc:276-378): the Gauss-Seidel sweep
against a row-by-row loop and against the wavefront order the device uses, GMRES against its defining minimisation
property, the whole solver against a sparse direct solve."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from oracle import refksp
from tests.test_oracle_solver import _problem


def _mg(orc, ex, ey, ez, nlv, kind="synth"):
    nx, ny, nz, h, KE, N, b, x, E = _problem(orc, ex, ey, ez, kind)
    mg = orc.MG(nx, ny, nz, 3, nlv)
    mg.assemble(KE, E, N)
    return mg, b, (nx, ny, nz)


def test_symmetric_sweep_equals_row_loops_and_wavefront_order(orc):
    mg, b, (nx, ny, nz) = _mg(orc, 8, 4, 4, 2)
    A = mg.csr(0)
    rng = np.random.default_rng(0)
    r = rng.standard_normal(A.shape[0])
    z = refksp.ssor_apply(A, r)
    y = refksp.gauss_seidel_sweep_loops(A, r, np.zeros_like(r))
    y = refksp.gauss_seidel_sweep_loops(A, r, y, backward=True)
    assert np.abs(z - y).max() <= 1e-13 * np.abs(z).max()
    # and z = (D + U)^-1 D (D + L)^-1 r
    import scipy.sparse as sp
    DL, DU = sp.tril(A, 0).tocsc(), sp.triu(A, 0).tocsc()
    w = spla.spsolve(DU, A.diagonal() * spla.spsolve(DL, r))
    assert np.abs(z - w).max() <= 1e-12 * np.abs(z).max()

    # the device runs a sweep wavefront by wavefront, t = i + 2 j + 4 k ascending (csrc/refksp.h): nodes of one t do not
    # couple and every lexicographically earlier neighbour has a smaller t -> the SAME numbers as the sequential sweep
    def wave_sweep(x, backward):
        A_ = A.tocsr()
        x = x.copy()
        nodes = [(i + 2 * j + 4 * k, i + nx * (j + ny * k)) for k in range(nz) for j in range(ny) for i in range(nx)]
        nodes.sort(key=lambda tn: tn[0], reverse=backward)   # within a wavefront: any order
        if backward:
            nodes = nodes[::1]
        for _, n in nodes:
            for c in (range(2, -1, -1) if backward else range(3)):
                i = 3 * n + c
                lo, hi = A_.indptr[i], A_.indptr[i + 1]
                cols, vals = A_.indices[lo:hi], A_.data[lo:hi]
                x[i] = x[i] + (r[i] - vals @ x[cols]) / vals[cols == i][0]
        return x
    yw = wave_sweep(np.zeros_like(r), False)
    ys = refksp.gauss_seidel_sweep_loops(A, r, np.zeros_like(r))
    assert np.array_equal(yw, ys)
    assert np.array_equal(wave_sweep(yw, True), refksp.gauss_seidel_sweep_loops(A, r, ys, backward=True))
    # the couplings the argument rests on: no matrix entry between two nodes of one wavefront
    coo = A.tocoo()
    t = lambda n: (n % nx) + 2 * ((n // nx) % ny) + 4 * (n // (nx * ny))
    rn, cn = coo.row // 3, coo.col // 3
    off = rn != cn
    assert (np.vectorize(t)(rn[off]) != np.vectorize(t)(cn[off])).all()


@pytest.mark.parametrize("pc", [0, 1])
def test_gmres_minimises_the_preconditioned_residual(orc, pc):
    mg, b, _ = _mg(orc, 16, 8, 8, 3)
    A = mg.csr(2)                               # 5 x 3 x 3 nodes
    n = A.shape[0]
    rng = np.random.default_rng(1)
    rhs = rng.standard_normal(n) * (A.diagonal() != 1.0)
    M = (lambda r: refksp.ssor_apply(A, r)) if pc else (lambda r: r / A.diagonal())
    x0 = rng.standard_normal(n) * 1e-3
    for k, guess in ((3, None), (4, x0)):
        x, its, hist = refksp.gmres_left(A, M, rhs, guess, k, k)
        assert its == k
        s = np.zeros(n) if guess is None else guess
        r0 = M(rhs - A @ s)
        K = [r0]
        for _ in range(k - 1):
            K.append(M(A @ K[-1]))
        K = np.array(K).T
        MAK = np.array([M(A @ K[:, q]) for q in range(k)]).T
        c = np.linalg.lstsq(MAK, r0, rcond=None)[0]
        best = np.linalg.norm(r0 - MAK @ c)
        mine = np.linalg.norm(M(rhs - A @ x))
        assert mine == pytest.approx(best, rel=1e-8)
        assert hist[-1] == pytest.approx(mine, rel=1e-8)      # the recurrence norm is the real one
    # restarts + convergence test
    m = 10 if pc else 50
    x, its, hist = refksp.gmres_left(A, M, rhs, None, m, 400, rtol=1e-10, test=True)
    assert m < its < 400 and hist[-1] <= 1e-10 * hist[0]
    assert np.abs(x - spla.spsolve(A.tocsc(), rhs)).max() <= 1e-7 * np.abs(x).max()


def test_reference_configuration_solves_the_cantilever(orc):
    mg, b, _ = _mg(orc, 16, 8, 8, 3)
    S = refksp.RefSolver(mg)                    # LinearElasticity.cc:620-746 as hard-coded
    x, its, hist = S.solve(b)
    assert 0 < its < 30
    bn = np.linalg.norm(b)
    assert hist[-1] <= 1e-5 * bn and (hist[:-1] > 1e-5 * bn).all()
    A = mg.csr(0)
    assert np.linalg.norm(b - A @ x) == pytest.approx(hist[-1], rel=1e-6)
    U = spla.spsolve(A.tocsc(), b)
    assert np.linalg.norm(x - U) <= 1e-3 * np.linalg.norm(U)
    # tight tolerance: the converged solution is the direct one (SURVEY 8(c) pin 5) ...
    S12 = refksp.RefSolver(mg, rtol=1e-12)
    x12, its12, _ = S12.solve(b)
    assert np.abs(x12 - U).max() <= 1e-9 * np.abs(U).max()
    # ... and a warm start from it needs no iteration
    assert S.solve(b, x0=x12)[1] == 0
    # far fewer outer iterations than CG + Chebyshev/Jacobi needs (the smoother is much stronger)
    assert its12 < mg.solve(b, rtol=1e-12)[1]


def test_reference_pdefilter_configuration(orc):
    ex, ey, ez = 16, 8, 8
    h = 1.0 / ey
    rmin = 2.56 * h
    kf, _ = orc.pde_kf(h, h, h, rmin / 2 / np.sqrt(3))
    mg = orc.MG(ex + 1, ey + 1, ez + 1, 1, 3)
    mg.assemble(kf)
    rng = np.random.default_rng(3)
    b = rng.random(mg.size(0)) * h ** 3
    S = refksp.RefSolver(mg, restart=20, rtol=1e-8, dtol=1e3, max_it=60, nsmooth=1, ncoarse=10, smooth_pc=0, coarse_pc=0,
                         coarse_restart=10)     # PDEFilter.cc:276-378
    x, its, hist = S.solve(b, x0=b / h ** 3)
    assert 0 < its < 60
    assert np.abs(x - spla.spsolve(mg.csr(0).tocsc(), b)).max() <= 1e-6 * np.abs(x).max()


def test_outer_restarts(orc):
    """FGMRES(3): the outer solver restarts every three iterations -- more iterations than FGMRES(100), the same solution"""
    mg, b, _ = _mg(orc, 16, 8, 8, 3)
    A = mg.csr(0)
    U = spla.spsolve(A.tocsc(), b)
    x100, its100, _ = refksp.RefSolver(mg, rtol=1e-10).solve(b)
    x3, its3, hist3 = refksp.RefSolver(mg, rtol=1e-10, restart=3).solve(b)
    assert its3 >= its100 > 3
    assert np.abs(x3 - U).max() <= 1e-8 * np.abs(U).max() and np.abs(x100 - U).max() <= 1e-8 * np.abs(U).max()
    assert np.linalg.norm(b - A @ x3) == pytest.approx(hist3[-1], rel=1e-5)

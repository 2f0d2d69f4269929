"""GPU parity of ksp_mode 1 -- the solver configuration the reference hard-codes (FGMRES + PCMG with GMRES / SOR level
solvers, LinearElasticity.cc:620-746; GMRES / Jacobi, PDEFilter.cc:276-378), csrc/refksp.h -- against its CPU
restatement oracle/refksp.py, building block by building block and as a whole.  FP64; tolerances per assertion."""
import numpy as np
import pytest

from oracle import refksp
from tests import scipy_check as sc
from tests.test_gpu_parity import dev, host, make, rel, tp  # noqa: F401  (tp is a fixture)

pytestmark = pytest.mark.gpu


def test_sor_and_jacobi_on_every_level(tp, orc):
    """PCSOR = one symmetric Gauss-Seidel sweep from a zero guess, run wavefront by wavefront on the device: the fine level
    takes its rows from the moduli (no stored matrix), the coarse levels from their stencils"""
    grid, le, mg, x, KE, N, R = make(tp, orc, 16, 8, 8, 3, ksp_mode=1)
    rng = np.random.default_rng(0)
    for l in range(3):
        A = mg.csr(l)
        r = rng.standard_normal(mg.size(l))
        assert rel(host(le.level_pc(l, 1, dev(r))), refksp.ssor_apply(A, r)) <= 1e-12
        assert rel(host(le.level_pc(l, 0, dev(r))), r / A.diagonal()) <= 1e-14


@pytest.mark.parametrize("pc", [1, 0])
def test_level_gmres(tp, orc, pc):
    grid, le, mg, x, KE, N, R = make(tp, orc, 16, 8, 8, 3, ksp_mode=1)
    rng = np.random.default_rng(1)
    for l in range(3):
        A = mg.csr(l)
        M = (lambda r: refksp.ssor_apply(A, r)) if pc else (lambda r: r / A.diagonal())
        b = rng.standard_normal(mg.size(l))
        # a smoother as PCMG runs it: 4 iterations, no test; from zero, then from the iterate
        xo, its_o, _ = refksp.gmres_left(A, M, b, None, 4, 4)
        xd, its = le.level_gmres(l, pc, 4, 4, dev(b), dev(np.zeros_like(b)), zero_guess=True)
        assert its == its_o == 4
        assert rel(host(xd), xo) <= 1e-10
        xo2, _, _ = refksp.gmres_left(A, M, b, xo, 4, 4)
        xd2, _ = le.level_gmres(l, pc, 4, 4, dev(b), xd.clone())
        assert rel(host(xd2), xo2) <= 1e-9
    # the coarse solve: restarts and the test on the preconditioned residual
    A = mg.csr(2)
    M = (lambda r: refksp.ssor_apply(A, r)) if pc else (lambda r: r / A.diagonal())
    xo, its_o, hist = refksp.gmres_left(A, M, b, None, 10, 60, rtol=1e-8, atol=1e-50, dtol=1e5, test=True)
    xd, its = le.level_gmres(2, pc, 10, 60, dev(b), dev(np.zeros_like(b)), zero_guess=True, rtol=1e-8)
    assert its == its_o
    assert rel(host(xd), xo) <= 1e-7


@pytest.mark.parametrize("kind,nlv", [("synth", 3), ("uniform", 2)])
def test_reference_configuration_vcycle_and_solve(tp, orc, kind, nlv):
    grid, le, mg, x, KE, N, R = make(tp, orc, 16, 8, 8, nlv, kind, ksp_mode=1)
    S = refksp.RefSolver(mg)
    r = np.random.default_rng(2).standard_normal(mg.n)
    assert rel(host(le.precond(dev(r))), S.vcycle(0, r)) <= 1e-8
    its = le.KSPSolve(hist_cap=300)
    Uo, its_o, hist_o = S.solve(R * N)
    assert its == its_o
    h = le.last_hist
    assert len(h) == len(hist_o)
    assert np.abs(h / hist_o - 1).max() <= 1e-6
    assert rel(host(le.U), Uo) <= 1e-7
    assert le.last_bnorm == pytest.approx(np.linalg.norm(R * N), rel=1e-14)
    assert le.KSPSolve() == 0      # warm start from the converged state
    # the option string says what ran
    s = le.petsc_options()
    assert "-ksp_type fgmres" in s and "-mg_levels_pc_type sor" in s and "-mg_coarse_ksp_gmres_restart 30" in s


def test_converged_state_is_the_fast_paths(tp, orc):
    """the two configurations solve the same system: at a tight tolerance displacement, compliance and sensitivities agree"""
    a = make(tp, orc, 32, 16, 16, 3, rtol=1e-11, max_it=400)
    b = make(tp, orc, 32, 16, 16, 3, rtol=1e-11, max_it=400, ksp_mode=1)
    out = []
    for grid, le, mg, x, KE, N, R in (a, b):
        dfdx, dgdx = grid.elem_vec(), grid.elem_vec()
        fx, gx = le.ComputeObjectiveConstraintsSensitivities(dfdx, dgdx, dev(x), 1e-9, 1.0, 3.0, 0.12)
        out.append((fx, host(dfdx), host(le.U), le.last_its))
    assert out[1][3] < out[0][3]                      # far stronger smoother: fewer outer iterations
    assert out[1][0] == pytest.approx(out[0][0], rel=1e-9)
    assert rel(out[1][1], out[0][1]) <= 1e-8
    assert rel(out[1][2], out[0][2]) <= 1e-8


def test_reference_pdefilter_configuration(tp, orc):
    ex, ey, ez = 16, 8, 8
    h = 1.0 / ey
    rmin = 2.56 * h
    grid = tp.Grid(ex + 1, ey + 1, ez + 1, h)
    f = tp.Filter(grid, 2, rmin, tp.SolverOptions.reference_pdefilter())
    x = np.random.default_rng(4).random(ex * ey * ez)
    xt, xp = grid.elem_vec(), grid.elem_vec()
    f.FilterProject(dev(x), xt, xp)
    kf, _ = orc.pde_kf(h, h, h, rmin / 2 / np.sqrt(3))
    mg = orc.MG(ex + 1, ey + 1, ez + 1, 1, 3)
    mg.assemble(kf)
    T = sc.elem_to_node_T(ex, ey, ez)
    S = refksp.RefSolver(mg, restart=20, rtol=1e-8, dtol=1e3, max_it=60, nsmooth=1, ncoarse=10, smooth_pc=0, coarse_pc=0,
                         coarse_restart=10)
    u, its_o, hist_o = S.solve(h ** 3 * (T @ x), x0=T @ x)      # PDEFilter.cc:198-210
    its, rn = f.last_pde_solve()
    assert its == its_o
    assert rn == pytest.approx(hist_o[-1], rel=1e-5)
    assert rel(host(xt), np.clip(T.T @ u, 0, 1)) <= 1e-9
    # and the fast configuration gives the same filtered field to the solver tolerance
    f0 = tp.Filter(grid, 2, rmin)
    xt0 = grid.elem_vec()
    f0.FilterProject(dev(x), xt0, xp)
    assert rel(host(xt), host(xt0)) <= 1e-6

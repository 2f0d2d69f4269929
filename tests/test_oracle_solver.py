"""Oracle pins that substitute for the reference's missing tests:
assembled == matrix-free, SPD + Dirichlet rows, Galerkin == explicit triple
product, converged quantities independent of the Krylov method (vs a sparse
direct solve), sensitivity vs finite differences."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from tests import scipy_check as sc


def _problem(orc, ex, ey, ez, kind="synth"):
    nx, ny, nz = ex + 1, ey + 1, ez + 1
    h = 1.0 / ey
    KE = orc.hex8_ke_box(h, h, h, 0.3)
    N, R = orc.cantilever_bc(nx, ny, nz, h)
    x = np.full(ex * ey * ez, 0.12) if kind == "uniform" else orc.synth_density(ex, ey, ez, h)
    E = orc.simp(x)
    return nx, ny, nz, h, KE, N, R * N, x, E


def test_cantilever_bc(orc):
    nx, ny, nz, h = 9, 5, 5, 0.25
    N, R = orc.cantilever_bc(nx, ny, nz, h)
    N3, R3 = N.reshape(nz, ny, nx, 3), R.reshape(nz, ny, nx, 3)
    assert (N3[:, :, 0, :] == 0).all() and (N3[:, :, 1:, :] == 1).all()
    assert (R3[..., :2] == 0).all()
    line = R3[0, :, nx - 1, 2]
    assert line[0] == -0.0005 and line[-1] == -0.0005 and (line[1:-1] == -0.001).all()
    assert np.count_nonzero(R) == ny


@pytest.mark.parametrize("kind", ["uniform", "synth"])
def test_assembled_equals_matrix_free(orc, kind):
    nx, ny, nz, h, KE, N, b, x, E = _problem(orc, 8, 4, 4, kind)
    mg = orc.MG(nx, ny, nz, 3, 1)
    mg.assemble(KE, E, N)
    rng = np.random.default_rng(0)
    u = rng.standard_normal(3 * nx * ny * nz)
    y1 = mg.apply(0, u)
    y2 = orc.matfree_apply(nx, ny, nz, 3, KE, E, N, u)
    assert np.abs(y1 - y2).max() <= 1e-13 * np.abs(y1).max()
    A = sc.assemble(8, 4, 4, KE, E, N)
    assert np.abs(A @ u - y1).max() <= 1e-13 * np.abs(y1).max()
    # Dirichlet rows return u_i; operator symmetric positive definite
    cl = N == 0
    assert np.array_equal(y1[cl], u[cl])
    Ao = mg.csr(0)
    assert abs(Ao - Ao.T).max() < 1e-15
    v = rng.standard_normal(u.size)
    assert v @ (Ao @ v) > 0


def test_galerkin_equals_triple_product(orc):
    nx, ny, nz, h, KE, N, b, x, E = _problem(orc, 16, 8, 8)
    mg = orc.MG(nx, ny, nz, 3, 3)
    mg.assemble(KE, E, N)
    A = sc.assemble(16, 8, 8, KE, E, N)
    cx, cy, cz = 16, 8, 8
    for l in range(1, 3):
        cx, cy, cz = cx // 2, cy // 2, cz // 2
        P = sc.interp3d(cx + 1, cy + 1, cz + 1, 3)
        A = (P.T @ A @ P).tocsr()
        Ao = mg.csr(l)
        assert Ao.shape == A.shape
        assert abs(Ao - A).max() <= 1e-14 * abs(A).max()
        # prolongation / restriction are P and P^T
        rng = np.random.default_rng(l)
        xc = rng.standard_normal(A.shape[0])
        assert np.allclose(mg.prolong(l - 1, xc), P @ xc, rtol=0, atol=1e-15 * 8)
        rf = rng.standard_normal(P.shape[0])
        assert np.allclose(mg.restrict(l - 1, rf), P.T @ rf, rtol=1e-14, atol=1e-14)
        # Lanczos estimate is a lower bound of, and close to, the true lambda_max(D^-1 A)
        d = A.diagonal()
        S = A.multiply(1 / np.sqrt(d)[:, None]).multiply(1 / np.sqrt(d)[None, :]).tocsr()
        lam = spla.eigsh(S, k=1, which="LA", return_eigenvectors=False)[0]
        assert 0.85 * lam <= mg.lam(l) <= lam * (1 + 1e-10)
    # fine level uses the rigorous element bound
    d = mg.diag(0)
    A0 = mg.csr(0)
    S = A0.multiply(1 / np.sqrt(d)[:, None]).multiply(1 / np.sqrt(d)[None, :]).tocsr()
    lam0 = spla.eigsh(S, k=1, which="LA", return_eigenvectors=False)[0]
    assert lam0 <= mg.lam(0) * (1 + 1e-12)


@pytest.mark.parametrize("kind", ["uniform", "synth"])
def test_converged_solution_is_solver_independent(orc, kind):
    """Pin (5): at rtol 1e-12 CG+MG, plain CG and a sparse direct solve agree on
    U, fx, dfdx to 1e-10 -- this is what backs the north star's 1e-10 claims."""
    ex, ey, ez = 16, 8, 8
    nx, ny, nz, h, KE, N, b, x, E = _problem(orc, ex, ey, ez, kind)
    mg = orc.MG(nx, ny, nz, 3, 3)
    mg.assemble(KE, E, N)
    U1, its1, hist1 = mg.solve(b, rtol=1e-13, maxit=400)
    assert 0 < its1 < 400
    assert hist1[-1] <= 1e-13 * np.linalg.norm(b)
    A = sc.assemble(ex, ey, ez, KE, E, N)
    U3 = spla.spsolve(A.tocsc(), b)
    scale = np.abs(U3).max()
    assert np.abs(U1 - U3).max() <= 1e-9 * scale
    f1, g1, df1, dg1 = orc.compliance_sens(nx, ny, nz, KE, U1, x)
    f3, g3, df3, dg3 = orc.compliance_sens(nx, ny, nz, KE, U3, x)
    assert abs(f1 - f3) <= 1e-10 * abs(f3)
    assert np.abs(df1 - df3).max() <= 1e-9 * np.abs(df3).max()
    # compliance identity  fx = U^T K U = b^T U
    assert abs(f3 - b @ U3) <= 1e-9 * abs(f3)
    if kind == "uniform":
        U2, its2, _ = mg.solve(b, rtol=1e-13, maxit=5000, use_pc=False)
        assert its2 > its1
        assert np.abs(U2 - U3).max() <= 1e-8 * scale
    assert g1 == pytest.approx(x.mean() - 0.12, abs=1e-13)
    assert (dg1 == 1.0 / x.size).all()


def test_pcg_history_and_warm_start(orc):
    nx, ny, nz, h, KE, N, b, x, E = _problem(orc, 16, 8, 8, "uniform")
    mg = orc.MG(nx, ny, nz, 3, 3)
    mg.assemble(KE, E, N)
    U, its, hist = mg.solve(b, rtol=1e-5)
    assert 0 < its < 30
    assert hist[0] == pytest.approx(np.linalg.norm(b), rel=1e-14)  # zero start: r0 = b
    assert hist[-1] <= 1e-5 * np.linalg.norm(b) < hist[-2]
    # the recorded norms are true residual norms
    assert np.linalg.norm(b - mg.apply(0, U)) == pytest.approx(hist[-1], rel=1e-6)
    # warm start from the converged state: 0 iterations (KSPSetInitialGuessNonzero)
    U2, its2, hist2 = mg.solve(b, x0=U, rtol=1e-5)
    assert its2 == 0 and np.array_equal(U, U2)
    # preconditioner is linear and symmetric (required for CG)
    rng = np.random.default_rng(3)
    r1, r2 = rng.standard_normal(b.size), rng.standard_normal(b.size)
    z1, z2 = mg.precond(r1), mg.precond(r2)
    assert np.allclose(mg.precond(r1 + 2 * r2), z1 + 2 * z2, rtol=1e-10, atol=1e-10 * np.abs(z1).max())
    assert r2 @ z1 == pytest.approx(r1 @ z2, rel=1e-9)
    assert r1 @ z1 > 0


def test_sensitivity_matches_finite_difference(orc):
    ex, ey, ez = 8, 4, 4
    nx, ny, nz, h, KE, N, b, x, E = _problem(orc, ex, ey, ez)
    x = np.clip(x, 0.05, 1.0)

    def f(xv):
        A = sc.assemble(ex, ey, ez, KE, orc.simp(xv), N)
        U = spla.spsolve(A.tocsc(), b)
        return orc.compliance_sens(nx, ny, nz, KE, U, xv)

    f0, _, df, _ = f(x)
    rng = np.random.default_rng(5)
    for e in rng.choice(x.size, 4, replace=False):
        d = 1e-6
        xp, xm = x.copy(), x.copy()
        xp[e] += d
        xm[e] -= d
        fd = (f(xp)[0] - f(xm)[0]) / (2 * d)
        assert fd == pytest.approx(df[e], rel=2e-5)


def test_mesh_must_be_coarsenable(orc):
    with pytest.raises(ValueError):  # TopOpt.cc:183-201
        orc.MG(11, 5, 5, 3, 3)


@pytest.mark.parametrize("k,zero_guess", [(4, True), (4, False), (1, True), (1, False), (30, True), (7, False)])
def test_chebyshev_equals_petsc_three_term_recurrence(orc, k, zero_guess):
    """The oracle (and the kernels) run the Chebyshev iteration in the direction-vector form; KSPSolve_Chebyshev of
    PETSc 3.11 is written as a three-term recurrence (scale = 2/(emax+emin), alpha = 1 - scale*emin, mu = 1/alpha,
    c_{k+1} = 2 mu c_k - c_{k-1}, omega = 2 c_k/(alpha c_{k+1}); first update without an operator application when the
    guess is zero; `-ksp_max_it k` = the first update plus k-1 recurrence steps).  Restated here in numpy from the
    published algorithm, with the Jacobi preconditioner and the window [0.1, 1.1] x lambda_max estimate: the same
    iterates to rounding -- the two forms are one method."""
    nx, ny, nz, h, KE, N, b, x, E = _problem(orc, 8, 4, 4, "synth")
    mg = orc.MG(nx, ny, nz, 3, 1)
    mg.assemble(KE, E, N)
    lam = mg.lam(0)
    emin, emax = 0.1 * lam, 1.1 * lam
    dinv = 1.0 / mg.diag(0)
    rng = np.random.default_rng(3)
    x0 = np.zeros_like(b) if zero_guess else rng.standard_normal(b.size) * 1e-3
    A = lambda v: mg.apply(0, v)
    # ---- PETSc's form
    scale = 2.0 / (emax + emin)
    alpha = 1.0 - scale * emin
    mu = 1.0 / alpha
    omegaprod = 2.0 / alpha
    c_km1, c_k = 1.0, mu
    p_km1 = x0.copy()
    r = b.copy() if zero_guess else b - A(p_km1)
    p_k = p_km1 + scale * (dinv * r)
    for _ in range(1, k):
        c_kp1 = 2.0 * mu * c_k - c_km1
        omega = omegaprod * c_k / c_kp1
        r = b - A(p_k)
        p_kp1 = (1.0 - omega) * p_km1 + omega * p_k + omega * scale * (dinv * r)
        p_km1, p_k = p_k, p_kp1
        c_km1, c_k = c_k, c_kp1
    xo = mg.smooth(0, b, x0.copy(), k, zero_guess)
    assert np.abs(xo - p_k).max() <= 1e-12 * np.abs(p_k).max()


@pytest.mark.parametrize("warm", [False, True])
def test_cg_equals_petsc_ksp_cg_restated(orc, warm):
    """KSPSolve_CG with -ksp_norm_type unpreconditioned and KSPConvergedDefault (reference norm ||b||, also for a non-zero
    initial guess) restated in numpy around the oracle's V-cycle as the preconditioner: the oracle's own CG gives the
    same iteration count and residual history."""
    nx, ny, nz, h, KE, N, b, x, E = _problem(orc, 16, 8, 8, "synth")
    mg = orc.MG(nx, ny, nz, 3, 3)
    mg.assemble(KE, E, N)
    rtol, atol, dtol, maxit = 1e-7, 1e-50, 1e5, 200
    x0 = np.zeros_like(b)
    if warm:
        x0, _, _ = mg.solve(b, rtol=1e-2)
    # ---- the restatement
    X = x0.copy()
    R = b - mg.apply(0, X) if warm else b.copy()
    bnorm = np.linalg.norm(b)
    ttol = max(rtol * bnorm, atol)
    hist = [np.linalg.norm(R)]
    its = 0
    if hist[0] > ttol:
        Z = mg.precond(R)
        beta = float(Z @ R)
        P = None
        for i in range(maxit):
            its = i + 1
            P = Z.copy() if i == 0 else Z + (beta / betaold) * P
            W = mg.apply(0, P)
            a = beta / float(P @ W)
            X += a * P
            R -= a * W
            betaold = beta
            dp = np.linalg.norm(R)
            hist.append(dp)
            if dp <= ttol or dp >= dtol * bnorm:
                break
            Z = mg.precond(R)
            beta = float(Z @ R)
    U, its_o, hist_o = mg.solve(b, x0=x0 if warm else None, rtol=rtol, atol=atol, dtol=dtol, maxit=maxit)
    assert its_o == its
    assert np.abs(np.asarray(hist_o[: its + 1]) / np.asarray(hist) - 1).max() <= 1e-8
    assert np.abs(U - X).max() <= 1e-9 * np.abs(X).max()


def _petsc_chebyshev(A, dinv, emin, emax, b, x0, k, zero_guess):
    """KSPSolve_Chebyshev of PETSc 3.11 (three-term form, PCJACOBI, fixed eigenvalues, k = max_it, no test), see
    test_chebyshev_equals_petsc_three_term_recurrence"""
    if k == 0:
        return x0.copy()
    scale = 2.0 / (emax + emin)
    alpha = 1.0 - scale * emin
    mu = 1.0 / alpha
    omegaprod = 2.0 / alpha
    c_km1, c_k = 1.0, mu
    p_km1 = x0.copy()
    r = b.copy() if zero_guess else b - A @ p_km1
    p_k = p_km1 + scale * (dinv * r)
    for _ in range(1, k):
        c_kp1 = 2.0 * mu * c_k - c_km1
        omega = omegaprod * c_k / c_kp1
        r = b - A @ p_k
        p_kp1 = (1.0 - omega) * p_km1 + omega * p_k + omega * scale * (dinv * r)
        p_km1, p_k = p_k, p_kp1
        c_km1, c_k = c_k, c_kp1
    return p_k


@pytest.mark.parametrize("nlv,ns,nc", [(3, 4, 30), (4, 2, 45), (2, 1, 10)])
def test_vcycle_equals_petsc_pcmg_restated(orc, nlv, ns, nc):
    """PCApply_MG of PETSc 3.11 -- PC_MG_MULTIPLICATIVE, one V-cycle, x zeroed at the top and on every coarser level
    (PCMGMCycle_Private), pre-smoother from the zero guess, default residual b - A x, restriction = transpose of the
    interpolation (only PCMGSetInterpolation is called, LinearElasticity.cc:704), coarse KSP, correction added, the SAME
    smoother again from the iterate -- restated in numpy on the oracle's level matrices and transfer operators, with
    PETSc's Chebyshev recurrence as smoother and coarse solver (the option string of SURVEY 8(d) with the oracle's
    numeric windows): equal to the oracle's own V-cycle to rounding."""
    nx, ny, nz, h, KE, N, b, x, E = _problem(orc, 32, 16, 16, "synth")
    mg = orc.MG(nx, ny, nz, 3, nlv, ns, nc)
    mg.assemble(KE, E, N)
    A = [mg.csr(l) for l in range(nlv)]
    dinv = [1.0 / mg.diag(l) for l in range(nlv)]

    def window(l):
        coarsest = l == nlv - 1 and l > 0
        return (mg.lam_min(l) if coarsest else 0.1 * mg.lam(l)), 1.1 * mg.lam(l)

    def cycle(l, rhs):
        lo, hi = window(l)
        if l == nlv - 1:
            return _petsc_chebyshev(A[l], dinv[l], lo, hi, rhs, np.zeros_like(rhs), nc, True)
        xl = _petsc_chebyshev(A[l], dinv[l], lo, hi, rhs, np.zeros_like(rhs), ns, True)
        rc = mg.restrict(l, rhs - A[l] @ xl)
        xl = xl + mg.prolong(l, cycle(l + 1, rc))
        return _petsc_chebyshev(A[l], dinv[l], lo, hi, rhs, xl, ns, False)

    r = np.random.default_rng(5).standard_normal(b.size) * N
    z = mg.precond(r)
    zr = cycle(0, r)
    assert np.abs(z - zr).max() <= 1e-11 * np.abs(zr).max()


def test_transfer_equals_dmda_q1_interpolation_restated(orc):
    """DMCreateInterpolation on a DMDA with Q1 elements and refinement factor 2 (what the reference passes to
    PCMGSetInterpolation, LinearElasticity.cc:703-705): coarse node I sits on fine node 2I, odd fine nodes average their
    two neighbours, dof by dof -- i.e. the Kronecker product Pz x Py x Px x I_3 of 1-D linear interpolations.  The
    oracle's prolongation is that matrix, its restriction the transpose (MatRestrict of PCMG without an own restriction)."""
    import scipy.sparse as sp
    nx, ny, nz, h, KE, N, b, x, E = _problem(orc, 16, 8, 12, "synth")
    mg = orc.MG(nx, ny, nz, 3, 3)
    mg.assemble(KE, E, N)

    def p1(nf):
        nc = (nf - 1) // 2 + 1
        P = sp.lil_matrix((nf, nc))
        for i in range(nf):
            if i % 2 == 0:
                P[i, i // 2] = 1.0
            else:
                P[i, (i - 1) // 2] = 0.5
                P[i, (i + 1) // 2] = 0.5
        return P.tocsr()

    rng = np.random.default_rng(9)
    dims = (nx, ny, nz)
    for l in range(2):
        fx, fy, fz = [(d - 1) // (1 << l) + 1 for d in dims]
        P = sp.kron(sp.kron(sp.kron(p1(fz), p1(fy)), p1(fx)), sp.identity(3)).tocsr()
        assert P.shape == (mg.size(l), mg.size(l + 1))
        xc = rng.standard_normal(mg.size(l + 1))
        rf = rng.standard_normal(mg.size(l))
        assert np.abs(mg.prolong(l, xc) - P @ xc).max() <= 1e-15 * np.abs(xc).max() * 8
        assert np.abs(mg.restrict(l, rf) - P.T @ rf).max() <= 1e-14 * np.abs(rf).max() * 27
        # Galerkin: the oracle's coarse matrix is P^T A P of ITS fine matrix (PCMGSetGalerkin both)
        Ac = (P.T @ mg.csr(l) @ P).tocsr()
        d = (Ac - mg.csr(l + 1)).tocoo()
        assert np.abs(d.data).max() <= 1e-12 * np.abs(Ac.data).max()


@pytest.mark.parametrize("cycles", [[2, 2, 2], [1, 2, 1], [1, 2, 2], [3, 1, 1]])
def test_w_cycles_equal_petsc_pcmg_restated(orc, cycles):
    """PCMGSetCycleType(W) / PCMGSetCycleTypeOnLevel: PCMGMCycle_Private cycles the next coarser level `cycles` times on
    the SAME right-hand side -- iterate zeroed once, the second cycle starts from the first one's result with the
    pre-smoother's non-zero-guess branch -- and once only into the coarsest level.  Restated in numpy with PETSc's
    Chebyshev recurrence; equal to the oracle's cycle.  The textbook form (new residual of the finer level, restricted
    again, next cycle from zero) is the same map because the coarse operators are Galerkin: checked too."""
    nx, ny, nz, h, KE, N, b, x, E = _problem(orc, 32, 16, 16, "synth")
    nlv, ns, nc = 4, 2, 10
    mg = orc.MG(nx, ny, nz, 3, nlv, ns, nc)
    mg.set_cycles(cycles)
    mg.assemble(KE, E, N)
    A = [mg.csr(l) for l in range(nlv)]
    dinv = [1.0 / mg.diag(l) for l in range(nlv)]
    win = lambda l: ((mg.lam_min(l) if l == nlv - 1 else 0.1 * mg.lam(l)), 1.1 * mg.lam(l))

    def petsc(l, rhs, x0, zero):
        lo, hi = win(l)
        if l == nlv - 1:
            return _petsc_chebyshev(A[l], dinv[l], lo, hi, rhs, x0, nc, zero)
        xl = _petsc_chebyshev(A[l], dinv[l], lo, hi, rhs, x0, ns, zero)
        rc = mg.restrict(l, rhs - A[l] @ xl)
        xc = np.zeros_like(rc)
        for c in range(1 if l + 1 == nlv - 1 else cycles[l]):
            xc = petsc(l + 1, rc, xc, c == 0)
        return _petsc_chebyshev(A[l], dinv[l], lo, hi, rhs, xl + mg.prolong(l, xc), ns, False)

    def textbook(l, rhs):
        lo, hi = win(l)
        if l == nlv - 1:
            return _petsc_chebyshev(A[l], dinv[l], lo, hi, rhs, np.zeros_like(rhs), nc, True)
        xl = _petsc_chebyshev(A[l], dinv[l], lo, hi, rhs, np.zeros_like(rhs), ns, True)
        for c in range(1 if l + 1 == nlv - 1 else cycles[l]):
            xl = xl + mg.prolong(l, textbook_inner(l + 1, mg.restrict(l, rhs - A[l] @ xl)))
        return _petsc_chebyshev(A[l], dinv[l], lo, hi, rhs, xl, ns, False)

    def textbook_inner(l, rhs):
        return textbook(l, rhs)

    r = np.random.default_rng(6).standard_normal(b.size) * N
    z = mg.precond(r)
    zp = petsc(0, r, np.zeros_like(r), True)
    assert np.abs(z - zp).max() <= 1e-11 * np.abs(zp).max()
    # the textbook W-cycle applies the WHOLE coarser cycle (with its own pre-smoothing from zero) to the new residual: the
    # same affine map only where the cycled level's iteration is stationary -- it is (fixed Chebyshev coefficients)
    zt = textbook(0, r)
    assert np.abs(z - zt).max() <= 1e-9 * np.abs(zt).max()
    # and it pays in Krylov iterations
    its_w = mg.solve(b, rtol=1e-8)[1]
    mg.set_cycles([1, 1, 1])
    assert its_w < mg.solve(b, rtol=1e-8)[1] or cycles == [3, 1, 1]


@pytest.mark.parametrize("nlv,cycles", [(3, (1, 1)), (4, (1, 2, 1))])
def test_exact_coarse_solve_equals_sparse_lu(orc, nlv, cycles):
    """MG.set_coarse_direct: the coarsest level solved exactly (banded Cholesky of the oracle, chol_band_factor /
    chol_band_solve; the reference's coarse KSP runs to rtol 1e-8, LinearElasticity.cc:628-632) -- the V-cycle restated in
    numpy with scipy's sparse LU as coarse solver gives the same preconditioner to rounding, and the exact solve is never
    worse than the 20-step Chebyshev run it replaces."""
    nx, ny, nz, h, KE, N, b, x, E = _problem(orc, 32, 16, 16, "synth")
    mg = orc.MG(nx, ny, nz, 3, nlv, 2, 20)
    mg.set_coarse_direct(True)
    mg.set_cycles(list(cycles))
    mg.assemble(KE, E, N)
    A = [mg.csr(l) for l in range(nlv)]
    dinv = [1.0 / mg.diag(l) for l in range(nlv)]
    lu = spla.splu(A[-1].tocsc())

    def cycle(l, rhs):   # V-cycle
        if l == nlv - 1:
            return lu.solve(rhs)
        lo, hi = 0.1 * mg.lam(l), 1.1 * mg.lam(l)
        xl = _petsc_chebyshev(A[l], dinv[l], lo, hi, rhs, np.zeros_like(rhs), 2, True)
        xl = xl + mg.prolong(l, cycle(l + 1, mg.restrict(l, rhs - A[l] @ xl)))
        return _petsc_chebyshev(A[l], dinv[l], lo, hi, rhs, xl, 2, False)

    r = np.random.default_rng(6).standard_normal(b.size) * N
    if all(c == 1 for c in cycles):
        z, zr = mg.precond(r), cycle(0, r)
        assert np.abs(z - zr).max() <= 1e-10 * np.abs(zr).max()
    U_d, its_d, hist_d = mg.solve(b, rtol=1e-6, maxit=300)
    mg2 = orc.MG(nx, ny, nz, 3, nlv, 2, 20)
    mg2.set_cycles(list(cycles))
    mg2.assemble(KE, E, N)
    U_c, its_c, hist_c = mg2.solve(b, rtol=1e-6, maxit=300)
    assert its_d <= its_c + 1 and np.abs(U_d - U_c).max() <= 1e-4 * np.abs(U_c).max()


def test_arbiter_is_the_same_algorithm_in_extended_precision():
    """oracle/arbiter.py: topopt_oracle.c rebuilt with `long double` for `double`.  Same inputs -> the same iteration count, a
    residual history and a compliance that agree with the double-precision oracle to rounding (1e-11 on a 10^4-DOF mesh), and
    arrays that really are 80-bit: its residuals keep falling below what double precision can represent relative to ||b||."""
    import numpy as np
    from oracle import arbiter as arb
    from oracle import oracle as orc
    assert arb.REAL is np.longdouble and orc.REAL is np.float64 and np.finfo(np.longdouble).nmant >= 63
    ex, ey, ez = 16, 8, 8
    nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
    x = orc.synth_density(ex, ey, ez, h)
    KE = orc.hex8_ke_box(h, h, h, 0.3)
    N, R = orc.cantilever_bc(nx, ny, nz, h)
    xp = orc.Filter(nx, ny, nz, h, 2.56 * h).project(1, x)[1]
    E = orc.simp(xp)
    out = {}
    for name, m in (("f64", orc), ("ld", arb)):
        mg = m.MG(nx, ny, nz, 3, 3, 2, 30)
        mg.assemble(KE, E, N)
        U, its, hist = mg.solve(m.f64(R * N), rtol=1e-5)
        Ut, its_t, hist_t = mg.solve(m.f64(R * N), rtol=1e-14, maxit=60)
        fx, gx, df, _ = m.compliance_sens(nx, ny, nz, KE, U, xp)
        out[name] = (its, np.asarray(hist, dtype=np.longdouble), fx, np.asarray(df, dtype=np.float64), hist_t, U.dtype)
    assert out["ld"][5] == np.longdouble and out["f64"][5] == np.float64
    assert out["f64"][0] == out["ld"][0]
    assert np.abs(out["f64"][1] / out["ld"][1] - 1).max() < 1e-11
    assert abs(out["f64"][2] / out["ld"][2] - 1) < 1e-12
    assert np.abs(out["f64"][3] - out["ld"][3]).max() <= 1e-12 * np.abs(out["ld"][3]).max()
    # extended precision: the true residual b - A x reaches 1e-14 ||b|| in the arbiter
    assert float(out["ld"][4][-1] / out["ld"][4][0]) <= 1e-14

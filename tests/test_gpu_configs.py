"""BASELINE.json configurations on the GPU.

Every config is exercised twice: (1) AS CONFIGURED (load case + filter type + multigrid depth) on a small mesh
against the CPU oracle, (2) at its FULL size through size-independent properties (symmetry / adjointness /
constant preservation / true-residual / compliance identity) -- the oracle cannot finish those sizes in seconds.
C3 and C5 are 8-GPU slab configurations: their slab geometry (8 ranks, 4 levels, 2 coarse layers per rank on the
coarsest distributed level, replicated coarsest level) runs as 8 ranks sharing one GPU on a mesh reduced in x-y.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def tp():
    import topopt_in_petsc_amd as tp
    tp.load_library()
    assert torch.cuda.is_available()
    return tp


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_product_kf_bits(tp):
    """The PRODUCT's Helmholtz element matrix (csrc/elements.h) against the numbers produced by running the
    reference's own PDEFilterMatrix (tests/golden/ref_kf.bin, make_ref_vectors.sh): bit for bit."""
    raw = np.fromfile(os.path.join(G, "ref_kf.bin")).reshape(-1, 4 + 64 + 8)
    assert len(raw) == 12
    for row in raw:
        dx, dy, dz, rmin = row[:4]
        grid = tp.Grid(9, 9, 9, (dx, dy, dz))
        f = tp.Filter(grid, 2, rmin)
        assert np.array_equal(f.KF(), row[4:68]), (dx, dy, dz, rmin)


def test_c4_as_configured_small_vs_oracle(tp, orc):
    """configs[3] = MBB load case AND the Helmholtz (PDE) filter in one design step, 3 levels (24x8x8 stand-in for
    192x64x64): filtered density, residual history, U, objective, sensitivities, filtered sensitivities."""
    ex, ey, ez, nlv = 24, 8, 8, 3
    nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
    grid = tp.Grid(nx, ny, nz, h)
    le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=nlv, rtol=1e-9, max_it=300))
    le.SetUpLoadAndBC_MBB()
    N, R = host(le.N), host(le.RHS)
    popt = dict(nlvls=3, rtol=1e-8, dtol=1e3, max_it=60, nsmooth=2, ncoarse=10)
    flt = tp.Filter(grid, 2, 2.56 * h, tp.SolverOptions(**popt))
    x = grid.synth_density()
    xt, xp, df, dg = grid.elem_vec(), grid.elem_vec(), grid.elem_vec(), grid.elem_vec()
    flt.FilterProject(x, xt, xp)                                                       # main.cc:98, Filter.cc:73-102
    its_f = flt.last_pde_solve()[0]
    fx, gx = le.ComputeObjectiveConstraintsSensitivities(df, dg, xp, 1e-9, 1.0, 3.0, 0.12, hist_cap=400)
    df_raw = df.clone()
    flt.Gradients(x, xt, df, [dg])                                                     # Filter.cc:195-199
    # ---- oracle, same N / RHS
    xo = orc.synth_density(ex, ey, ez, h)
    opf = orc.PDEFilter(nx, ny, nz, h, 2.56 * h, nlv=3, nsmooth=2, ncoarse=10)
    xto, its_o, _ = opf.apply(xo)
    xto = np.clip(xto, 0.0, 1.0)
    assert its_f == its_o
    assert rel(host(xt), xto) <= 1e-9 and rel(host(xp), xto) <= 1e-9
    KE = orc.hex8_ke_box(h, h, h, 0.3)
    mg = orc.MG(nx, ny, nz, 3, nlv)
    mg.assemble(KE, orc.simp(xto), N)
    U, its, hist = mg.solve(R * N, rtol=1e-9, maxit=300)
    assert le.last_its == its
    k = min(10, its)
    assert np.abs(le.last_hist[:k] / hist[:k] - 1).max() <= 1e-8     # the filtered densities differ at 1e-9
    assert rel(host(le.U), U) <= 1e-7
    fo, go, dfo, dgo = orc.compliance_sens(nx, ny, nz, KE, U, xto)
    assert abs(fx / fo - 1) <= 1e-8 and abs(gx - go) <= 1e-10
    assert rel(host(df_raw), dfo) <= 1e-7
    dfo_f, _, _ = opf.apply(dfo)
    dgo_f, _, _ = opf.apply(dgo)
    assert rel(host(df), dfo_f) <= 1e-7
    assert rel(host(dg), dgo_f) <= 1e-7


CONFIGS = {
    # name: (ex, ey, ez, nlvls, filter type, load case)
    "C2_128x64x64": (128, 64, 64, 3, 1, "cantilever"),
    "C4_192x64x64_mbb_pde": (192, 64, 64, 3, 2, "mbb"),
    "C3_256x128x128_one_gpu": (256, 128, 128, 4, 1, "cantilever"),
    "metric_128cubed": (128, 128, 128, 4, 1, "cantilever"),
    "C5_512x256x256_one_gpu": (512, 256, 256, 4, 1, "cantilever"),   # 101.6 M DOF, ~35 GB of the 288 GB
    # the parameterisation bench.py actually runs (WORKLOADS): depth, step counts, W-cycles on the middle levels
    "metric_128cubed_bench_cycle": (128, 128, 128, 5, 1, "cantilever", dict(nsmooth=2, ncoarse=20, coarse_direct=1), [1, 3, 1, 1]),
    "C3_256x128x128_bench_cycle": (256, 128, 128, 6, 1, "cantilever", dict(nsmooth=2, ncoarse=20, coarse_direct=1), [1, 3, 1, 1, 1]),
    # round 2's bench cycle (W on levels 2-3, Chebyshev coarse run)
    "metric_128cubed_round2_cycle": (128, 128, 128, 5, 1, "cantilever", dict(nsmooth=2, ncoarse=20), [1, 2, 2, 1]),
}


def test_c2_full_size_against_the_oracle(tp, orc):
    """BASELINE configs[1] AT FULL SIZE (128 x 64 x 64 elements, 1 635 075 DOF, "matrix-free PCG + 3-level GMG") against the
    oracle's assembled-CSR solve on the reference's KE: iteration count, every ||r_k||, compliance, volume and filtered
    sensitivities -- north_star's 1e-10 (history, fx), both at the bench's rtol 1e-5 from the cold start.  The workload of
    `bench.py --workload c2` (3 levels, Chebyshev(2), 45 coarse steps); VERDICT r5 weak 4: full-size comparisons existed for the
    metric mesh and C4 only."""
    ex, ey, ez, nlv = 128, 64, 64, 3
    nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
    grid = tp.Grid(nx, ny, nz, h)
    le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=nlv, rtol=1e-5, nsmooth=2, ncoarse=45))
    le.SetUpLoadAndBC()
    flt = tp.Filter(grid, 1, 2.56 * h)
    x = grid.synth_density(12345)
    xt, xp, df, dg = grid.elem_vec(), grid.elem_vec(), grid.elem_vec(), grid.elem_vec()
    flt.FilterProject(x, xt, xp)
    fx, gx = le.ComputeObjectiveConstraintsSensitivities(df, dg, xp, 1e-9, 1.0, 3.0, 0.12, hist_cap=256)
    flt.Gradients(x, xt, df, [dg])
    xo = orc.synth_density(ex, ey, ez, h)
    of = orc.Filter(nx, ny, nz, h, 2.56 * h)
    xto, xpo = of.project(1, xo)
    KE = orc.hex8_ke_box(h, h, h, 0.3)
    N, R = orc.cantilever_bc(nx, ny, nz, h)
    mg = orc.MG(nx, ny, nz, 3, nlv, 2, 45)
    mg.assemble(KE, orc.simp(xpo), N)
    U, its, hist = mg.solve(R * N, rtol=1e-5)
    fo, go, dfo, dgo = orc.compliance_sens(nx, ny, nz, KE, U, xpo)
    dfo = of.gradient(1, xo, xto, dfo)
    assert le.last_its == its and 10 < its < 100
    assert np.abs(np.array(le.last_hist) / hist - 1).max() <= 1e-10
    assert abs(fx / fo - 1) <= 1e-10 and abs(gx - go) <= 1e-13
    assert np.abs(df.cpu().numpy() - dfo).max() <= 1e-9 * np.abs(dfo).max()
    assert np.abs(le.U.cpu().numpy() - U).max() <= 1e-9 * np.abs(U).max()
    grid.close()


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_config_full_size_properties(tp, name):
    ex, ey, ez, nlv, ftype, bc = CONFIGS[name][:6]
    kw, cycles = (CONFIGS[name][6], CONFIGS[name][7]) if len(CONFIGS[name]) > 6 else ({}, None)
    h = 1.0 / ey
    grid = tp.Grid(ex + 1, ey + 1, ez + 1, h)
    le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=nlv, rtol=1e-8, max_it=400, **kw))
    if cycles:
        le.set_cycles(cycles)
    popt = tp.SolverOptions(nlvls=3, rtol=1e-10, dtol=1e3, max_it=100, nsmooth=2, ncoarse=10) if ftype == 2 else None
    flt = tp.Filter(grid, ftype, 2.56 * h, popt)
    le.SetUpLoadAndBC_MBB() if bc == "mbb" else le.SetUpLoadAndBC()
    x = grid.synth_density()
    xt, xp = grid.elem_vec(), grid.elem_vec()
    flt.FilterProject(x, xt, xp)
    ftol = 1e-12 if ftype == 1 else 1e-7   # PDE filter: an iterative solve sits inside
    # filter: preserves constants; adjoint identity <F x, y> = <x, F^T y>
    one, o1, o2 = grid.elem_vec(1.0), grid.elem_vec(), grid.elem_vec()
    flt.FilterProject(one, o1, o2)
    assert float((o1 - 1).abs().max()) < (1e-14 if ftype == 1 else 1e-7)
    if ftype == 1:
        assert float(xt.min()) >= float(x.min()) - 1e-15 and float(xt.max()) <= float(x.max()) + 1e-15
    else:
        assert float(xt.min()) >= 0.0 and float(xt.max()) <= 1.0                      # Filter.cc:81-100 clamp
    y = torch.rand_like(x)
    g = y.clone()
    flt.Gradients(x, xt, g, [])
    xt_unclamped = xt
    if ftype == 2:   # the adjoint identity holds for the linear filter, before the clamp of FilterProject
        xt_unclamped = grid.elem_vec()
        flt.Gradients(x, xt, xt_unclamped.copy_(x), [])
    assert float(torch.dot(xt_unclamped, y)) == pytest.approx(float(torch.dot(x, g)), rel=ftol)
    # operator: symmetric, Dirichlet rows, positive
    le.AssembleStiffnessMatrix(xp, 1e-9, 1.0, 3.0)
    u, v = torch.randn_like(le.U), torch.randn_like(le.U)
    Au, Av = le.MatMult(u), le.MatMult(v)
    assert float(torch.dot(v, Au)) == pytest.approx(float(torch.dot(u, Av)), rel=1e-11)
    cl = le.N == 0
    assert int(cl.sum()) > 0 and torch.equal(Au[cl], u[cl])
    assert float(torch.dot(u, Au)) > 0
    # every level operator is symmetric; preconditioner symmetric positive (CG requirement)
    for l in range(1, nlv):
        n_l = 3 * le.level_nodes(l)
        a, b = torch.randn(n_l, dtype=torch.float64, device="cuda"), torch.randn(n_l, dtype=torch.float64, device="cuda")
        assert float(torch.dot(b, le.level_apply(l, a))) == pytest.approx(float(torch.dot(a, le.level_apply(l, b))), rel=1e-10)
    zu, zv = le.precond(u), le.precond(v)
    assert float(torch.dot(v, zu)) == pytest.approx(float(torch.dot(u, zv)), rel=1e-9)
    assert float(torch.dot(u, zu)) > 0
    # solve: true residual matches the reported one; compliance identity fx = b^T U; monotone energy-norm CG
    its = le.KSPSolve(hist_cap=512)
    assert 0 < its < 400
    b = le.RHS * le.N
    r = b - le.MatMult(le.U)
    assert float(r.norm()) == pytest.approx(le.last_rnorm, rel=1e-4)
    assert le.last_rnorm <= 1e-8 * le.last_bnorm
    fx, gx = le.Objective(xp, 1e-9, 1.0, 3.0, 0.12)
    assert fx == pytest.approx(float(torch.dot(b, le.U)), rel=1e-7)
    assert gx == pytest.approx(float(xp.mean()) - 0.12, abs=1e-12)
    # the option string a PETSc user would paste (numeric Chebyshev windows of this assembly)
    opts = le.petsc_options()
    assert "-ksp_type cg" in opts and "-pc_mg_levels %d" % nlv in opts and "-mg_coarse_ksp_chebyshev_eigenvalues" in opts
    for k in range(1, nlv):
        assert "-mg_levels_%d_ksp_chebyshev_eigenvalues" % k in opts


def test_petsc_option_string_numbers(tp, orc):
    """tp_elasticity_petsc_options carries exactly the windows the solver uses (and the oracle computes)."""
    import re
    ex, ey, ez, nlv = 16, 8, 8, 3
    nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
    grid = tp.Grid(nx, ny, nz, h)
    le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=nlv))
    le.SetUpLoadAndBC()
    x = orc.synth_density(ex, ey, ez, h)
    le.AssembleStiffnessMatrix(dev(x), 1e-9, 1.0, 3.0)
    N, R = orc.cantilever_bc(nx, ny, nz, h)
    mg = orc.MG(nx, ny, nz, 3, nlv)
    mg.assemble(orc.hex8_ke_box(h, h, h, 0.3), orc.simp(x), N)
    opts = le.petsc_options()
    for l in range(nlv):
        k = nlv - 1 - l                      # PETSc numbers levels from the coarsest (0) to the finest
        pre = "mg_coarse" if k == 0 else "mg_levels_%d" % k
        m = re.search(r"-%s_ksp_chebyshev_eigenvalues ([-0-9.e+]+),([-0-9.e+]+)" % pre, opts)
        assert m, (pre, opts)
        lo, hi = float(m.group(1)), float(m.group(2))
        assert hi == pytest.approx(1.1 * mg.lam(l), rel=1e-9)
        if k > 0:
            assert lo == pytest.approx(0.1 * mg.lam(l), rel=1e-9)
            assert "-%s_ksp_max_it 4" % pre in opts
        else:
            assert 0 < lo < 0.1 * hi and "-mg_coarse_ksp_max_it 30" in opts
    assert "-pc_mg_galerkin both" in opts and "-ksp_norm_type unpreconditioned" in opts


@pytest.mark.gpu
def test_graphed_coarsest_smoothing_bits(tp):
    """TP_SMOOTH_GRAPH=1: the 30 coarsest-level Chebyshev steps of a V-cycle replayed from a hipGraph (captured on a
    stream of the library's own, replayed on the grid's stream, two role-alternating variants) -- same bits as the
    direct launches, fewer launches."""
    ex, ey, ez, nl = 64, 32, 32, 4
    g = tp.Grid(ex + 1, ey + 1, ez + 1, 1.0 / ey)
    le = tp.LinearElasticity(g, tp.SolverOptions(nlvls=nl))
    le.SetUpLoadAndBC()
    res = []
    for graph in (False, True, False):
        if graph:
            os.environ["TP_SMOOTH_GRAPH"] = "1"
        else:
            os.environ.pop("TP_SMOOTH_GRAPH", None)
        os.environ["TP_NO_COARSE_XCD"] = "1"   # the comparison is with the separate launches, not with the one-XCD run
        try:
            runs = []
            for seed in (1, 2):   # a second design: the Chebyshev windows change, the graphs are captured again
                le.AssembleStiffnessMatrix(g.synth_density(seed), 1e-9, 1.0, 3.0)
                le.pop_stats()
                le.U.zero_()              # cold start: the same initial guess in every mode
                le.KSPSolve()
                runs.append((host(le.U), le.last_its, le.pop_stats()[2]))
            res.append(runs)
        finally:
            os.environ.pop("TP_SMOOTH_GRAPH", None)
            os.environ.pop("TP_NO_COARSE_XCD", None)
    for k in range(2):
        assert np.array_equal(res[0][k][0], res[1][k][0]) and np.array_equal(res[0][k][0], res[2][k][0])
        assert res[0][k][1] == res[1][k][1] and res[0][k][1] > 4
        if not os.environ.get("TP_DEBUG_SYNC"):   # (per-launch synchronisation switches graph replay off)
            assert res[1][k][2] < res[0][k][2], (res[1][k][2], res[0][k][2])


@pytest.mark.gpu
@pytest.mark.parametrize("mesh,nl,single", [((64, 32, 32), 4, False), ((32, 32, 32), 5, True), ((48, 24, 24), 3, False),
                                            ((48, 24, 24), 4, True), ((64, 64, 64), 5, True), ((64, 64, 128), 4, False),
                                            ((64, 64, 256), 4, False)])
def test_coarsest_run_in_one_launch_bits(tp, mesh, nl, single):
    """csrc/coarse_run.h: the Chebyshev steps of the coarsest level as iterations inside ONE kernel.  Coarsest grids of
    <= 448 rows (7 x 4 x 4, 5^3, 3^3 nodes here) run in ONE workgroup with the iterate in LDS -- the default; larger ones
    (9 x 5 x 5, 13 x 7 x 7 nodes; 9 x 9 x 17 and 9 x 9 x 33 with 4 and 8 rows per thread) across workgroups with a barrier
    per step: by default (round 3) workgroups that have gathered on ONE XCD and exchange the iterate through its L2, opt-in
    (TP_COARSE_RUN=1) workgroups anywhere.  Either way:
    the same bits as the separate launches (TP_NO_COARSE_RUN=1), fewer launches."""
    ex, ey, ez = mesh
    g = tp.Grid(ex + 1, ey + 1, ez + 1, 1.0 / ey)
    le = tp.LinearElasticity(g, tp.SolverOptions(nlvls=nl, nsmooth=2, ncoarse=45))
    le.SetUpLoadAndBC()
    res = {}
    for mode, env in (("launches", {"TP_NO_COARSE_RUN": "1"}), ("default", {}), ("run", {"TP_COARSE_RUN": "1"}), ("no_xcd", {"TP_NO_COARSE_XCD": "1"})):
        os.environ.update(env)
        try:
            le.AssembleStiffnessMatrix(g.synth_density(3), 1e-9, 1.0, 3.0)
            le.pop_stats()
            le.U.zero_()
            le.KSPSolve()
            res[mode] = (host(le.U), le.last_its, le.pop_stats()[2])
        finally:
            for k in env:
                os.environ.pop(k, None)
    for mode in ("default", "run", "no_xcd"):
        assert np.array_equal(res["launches"][0], res[mode][0]) and res["launches"][1] == res[mode][1] > 4, mode
    assert res["run"][2] < 0.6 * res["launches"][2], (res["run"][2], res["launches"][2])
    assert res["default"][2] < 0.6 * res["launches"][2], (res["default"][2], res["launches"][2])
    assert (res["no_xcd"][2] < 0.6 * res["launches"][2]) == single, (res["no_xcd"][2], res["launches"][2])


@pytest.mark.gpu
@pytest.mark.parametrize("mesh,nl", [((64, 32, 32), 4), ((48, 24, 24), 3), ((64, 64, 64), 4), ((40, 40, 24), 3), ((64, 64, 128), 4)])
def test_coarsest_spectrum_in_one_launch(tp, mesh, nl):
    """csrc/coarse_run.h, k_lanczos_run_xcd: the 40 Lanczos steps of the coarsest level (full reorthogonalisation) inside ONE
    kernel on one XCD.  Row arithmetic as in the chain of launches, dot products summed in another order: both ends of the
    coarse Chebyshev window agree to rounding, the solve takes the same iterations; 280 launches less per set-up."""
    ex, ey, ez = mesh
    g = tp.Grid(ex + 1, ey + 1, ez + 1, 1.0 / ey)
    le = tp.LinearElasticity(g, tp.SolverOptions(nlvls=nl, nsmooth=2, ncoarse=30))
    le.SetUpLoadAndBC()
    res = {}
    for mode, env in (("launches", {"TP_NO_LANCZOS_XCD": "1"}), ("default", {})):
        os.environ.update(env)
        try:
            le.pop_stats()
            for _ in range(3):   # the control block must come back clean
                le.AssembleStiffnessMatrix(g.synth_density(3), 1e-9, 1.0, 3.0)
            n_setup = le.pop_stats()[2] / 3.0
            le.U.zero_()
            le.KSPSolve()
            res[mode] = (host(le.U), le.last_its, n_setup, le.level_lambda(nl - 1), le.level_lambda_min(nl - 1))
        finally:
            for k in env:
                os.environ.pop(k, None)
    a, b = res["launches"], res["default"]
    assert abs(b[3] / a[3] - 1) <= 1e-12 and abs(b[4] / a[4] - 1) <= 1e-11, (a[3:], b[3:])
    assert 0 < b[4] < b[3]
    assert a[1] == b[1] > 4 and np.abs(a[0] - b[0]).max() <= 1e-9 * np.abs(a[0]).max()
    assert b[2] < a[2], (a[2], b[2])

"""Regression against the frozen oracle outputs in tests/golden (made by make_oracle_golden.py):
CPU: the oracle still reproduces them; GPU: the HIP path matches the files."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


def test_oracle_reproduces_golden(orc):
    g = _load("oracle_16x8x8.npz")
    ex, ey, ez, nlv = [int(v) for v in g["dims"]]
    nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
    x = orc.synth_density(ex, ey, ez, h)
    assert np.array_equal(x, g["x"])
    flt = orc.Filter(nx, ny, nz, h, 2.56 * h)
    xt, xp = flt.project(1, x)
    assert np.allclose(xt, g["xTilde"], rtol=1e-15, atol=0) and np.allclose(flt.hs(), g["Hs"], rtol=1e-15)
    KE = orc.hex8_ke_box(h, h, h, 0.3)
    N, R = orc.cantilever_bc(nx, ny, nz, h)
    mg = orc.MG(nx, ny, nz, 3, nlv)
    mg.assemble(KE, orc.simp(xp), N)
    assert np.allclose(mg.apply(0, g["v"]), g["Kv"], rtol=1e-14, atol=1e-16)
    assert np.allclose([mg.lam(l) for l in range(nlv)], g["lam"], rtol=1e-11)
    assert mg.lam_min(nlv - 1) == pytest.approx(float(g["lam_min"]), rel=1e-9)
    U, its, hist = mg.solve(R * N, rtol=float(g["rtol"]), maxit=300)
    assert its == int(g["its"]) and np.allclose(hist, g["hist"], rtol=1e-8)
    fx, gx, df, dg = orc.compliance_sens(nx, ny, nz, KE, U, xp)
    assert fx == pytest.approx(float(g["fx"]), rel=1e-10) and np.allclose(df, g["dfdx"], rtol=1e-7, atol=1e-12)


def test_c1_scalars(orc):
    """BASELINE config C1 (48x24x24, 4 levels): iteration-1-like scalars"""
    g = _load("oracle_c1_scalars.npz")
    assert [int(v) for v in g["dims"]] == [48, 24, 24, 4]
    assert 0 < int(g["its"]) < 60 and g["hist"][-1] <= 1e-5 * g["hist"][0] * 1.0001


@pytest.mark.gpu
def test_gpu_matches_golden_files():
    import torch
    import topopt_in_petsc_amd as tp
    for name in ("oracle_16x8x8.npz", "oracle_c1_scalars.npz"):
        g = _load(name)
        ex, ey, ez, nlv = [int(v) for v in g["dims"]]
        rtol = float(g["rtol"]) if "rtol" in g else 1e-5
        nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
        grid = tp.Grid(nx, ny, nz, h)
        le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=nlv, rtol=rtol, max_it=300))
        le.SetUpLoadAndBC()
        flt = tp.Filter(grid, 1, 2.56 * h)
        x = grid.synth_density()
        xt, xp, df, dg = grid.elem_vec(), grid.elem_vec(), grid.elem_vec(), grid.elem_vec()
        flt.FilterProject(x, xt, xp)
        fx, gx = le.ComputeObjectiveConstraintsSensitivities(df, dg, xp, 1e-9, 1.0, 3.0, 0.12, hist_cap=400)
        assert le.last_its == int(g["its"])
        assert np.allclose(le.last_hist, g["hist"], rtol=1e-7)
        assert fx == pytest.approx(float(g["fx"]), rel=1e-8) and gx == pytest.approx(float(g["gx"]), abs=1e-13)
        assert np.allclose([le.level_lambda(l) for l in range(nlv)], g["lam"], rtol=1e-9)
        if "dfdx" in g:
            assert np.abs(x.cpu().numpy() - g["x"]).max() < 1e-15   # device sin() vs libm: last bits
            assert np.allclose(xt.cpu().numpy(), g["xTilde"], rtol=1e-13)
            assert np.allclose(df.cpu().numpy(), g["dfdx"], rtol=1e-6, atol=1e-11)
            v = torch.from_numpy(g["v"]).cuda()
            assert np.allclose(le.MatMult(v).cpu().numpy(), g["Kv"], rtol=1e-12, atol=1e-14)
            flt.Gradients(x, xt, df, [])
            assert np.allclose(df.cpu().numpy(), g["dfdx_filtered"], rtol=1e-6, atol=1e-11)
            pf = tp.Filter(grid, 2, 2.56 * h, tp.SolverOptions(nlvls=min(nlv, 3), rtol=1e-8, dtol=1e3, max_it=60,
                                                               nsmooth=2, ncoarse=10))
            pf.FilterProject(x, xt, xp)
            assert pf.last_pde_solve()[0] == int(g["its_pde"])
            assert np.allclose(xt.cpu().numpy(), np.clip(g["xpde"], 0, 1), rtol=1e-8, atol=1e-10)

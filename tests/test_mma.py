"""MMA: the oracle restatement on CPU (properties), and the device implementation
against it (GPU)."""
import os
import numpy as np
import pytest


def _toy(n=2000, seed=0):
    """a separable convex problem: min sum c_i / x_i  s.t. mean(x) <= v"""
    rng = np.random.default_rng(seed)
    c = rng.random(n) + 0.1
    return c, 0.3


def test_oracle_mma_converges_to_kkt_point(orc):
    c, v = _toy()
    n = c.size
    x = np.full(n, v)
    mma = orc.MMA(x, 1)
    xold = x.copy()
    f_hist = []
    for it in range(40):
        f = (c / x).sum()
        df = -c / x ** 2
        g = x.mean() - v
        dg = np.full(n, 1.0 / n)
        scale = 10.0 / f_hist[0] if f_hist else 10.0 / f
        f_hist.append(f)
        xmin, xmax = mma.SetOuterMovelimit(1e-3, 1.0, 0.2, x)
        x = mma.Update(x, df * scale, [g], [dg], xmin, xmax)
        assert (x >= xmin - 1e-15).all() and (x <= xmax + 1e-15).all()
        ch = mma.DesignChange(x, xold)
        if ch < 1e-4:
            break
    assert f_hist[-1] < f_hist[0]
    assert abs(x.mean() - v) < 1e-6                      # constraint active
    # analytic optimum: x_i proportional to sqrt(c_i) (clipped), mean v
    xs = np.sqrt(c)
    xs *= v / xs.mean()
    assert np.abs(x - xs).max() < 5e-3
    lam, z = mma.state()
    assert lam[0] > 0 and z == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("device_order", [False, True])
def test_device_mma_matches_oracle(orc, device_order):
    """device_order = False: the oracle sums left to right and cubes with pow() like MMA.cc -> agreement to 1e-12.
    device_order = True: the oracle performs the SAME operations in the SAME order as the HIP kernels (workgroup /
    shuffle-tree sums, cube by multiplication) -> the design update is bit-identical, iteration after iteration."""
    import torch
    import topopt_in_petsc_amd as tp
    c, v = _toy(12 * 8 * 8)
    n = c.size
    grid = tp.Grid(13, 9, 9, 0.125)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    x = np.full(n, v)
    xd = dev(x)
    m_o, m_d = orc.MMA(x, 1), tp.MMA(grid, xd, 1)
    if device_order:
        m_o.set_device_order()
    xold_o, xold_d = x.copy(), dev(x)
    xmin_d, xmax_d = grid.elem_vec(), grid.elem_vec()
    for it in range(8):
        f = (c / x).sum()
        df = -c / x ** 2 * (10.0 / 2000.0)
        g = x.mean() - v
        dg = np.full(n, 1.0 / n)
        xmin, xmax = m_o.SetOuterMovelimit(1e-3, 1.0, 0.2, x)
        m_d.SetOuterMovelimit(1e-3, 1.0, 0.2, xd, xmin_d, xmax_d)
        assert np.array_equal(xmin_d.cpu().numpy(), xmin) and np.array_equal(xmax_d.cpu().numpy(), xmax)
        x = m_o.Update(x, df, [g], [dg], xmin, xmax)
        m_d.Update(xd, dev(df), [g], [dev(dg)], xmin_d, xmax_d)
        assert m_d.last_inner == m_o.last_inner
        xg = xd.cpu().numpy()
        if device_order:
            assert np.array_equal(xg, x), (it, np.abs(xg - x).max())
            assert m_d.state()[0][0] == m_o.state()[0][0]
        assert np.abs(xg - x).max() <= 1e-12, (it, np.abs(xg - x).max())
        assert m_d.state()[0][0] == pytest.approx(m_o.state()[0][0], rel=1e-11)
        ch_o = m_o.DesignChange(x, xold_o)
        assert m_d.DesignChange(xd, xold_d) == pytest.approx(ch_o, abs=1e-12)
        xd.copy_(dev(x))  # identical inputs for the next step


@pytest.mark.gpu
def test_optimisation_loop_matches_oracle_loop(orc):
    """main.cc's loop on the device vs the same loop driven by the oracle, 48x24x24-like small mesh"""
    import topopt_in_petsc_amd as tp
    ex, ey, ez, nlv = 32, 16, 16, 3
    nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
    rmin = 2.56 * h
    opt = tp.TopOpt(nxyz=(nx, ny, nz), xc=(0, 2, 0, 1, 0, 1), nlvls=nlv, rmin=rmin,
                    solver=tp.SolverOptions(nlvls=nlv, rtol=1e-8))
    # ---- oracle loop
    KE = orc.hex8_ke_box(h, h, h, 0.3)
    N, R = orc.cantilever_bc(nx, ny, nz, h)
    flt = orc.Filter(nx, ny, nz, h, rmin)
    mg = orc.MG(nx, ny, nz, 3, nlv)
    n = ex * ey * ez
    x = np.full(n, 0.12)
    xt, xp = flt.project(1, x)
    mma = orc.MMA(x, 1)
    xold = x.copy()
    U = np.zeros(3 * nx * ny * nz)
    fscale = None
    for it in range(6):
        rec = opt.step()
        mg.assemble(KE, orc.simp(xp), N)
        U, its, hist = mg.solve(R * N, x0=U, rtol=1e-8)
        fx, gx, df, dg = orc.compliance_sens(nx, ny, nz, KE, U, xp)
        if fscale is None:
            fscale = 10.0 / fx
        df = flt.gradient(1, x, xt, df * fscale)
        dg = flt.gradient(1, x, xt, dg)
        xmin, xmax = mma.SetOuterMovelimit(0.0, 1.0, 0.2, x)
        x = mma.Update(x, df, [gx], [dg], xmin, xmax)
        ch = mma.DesignChange(x, xold)
        xt, xp = flt.project(1, x)
        assert rec["ksp_its"] == its, (it, rec["ksp_its"], its)
        assert rec["fx"] == pytest.approx(fx, rel=1e-7)
        assert rec["gx"] == pytest.approx(gx, abs=1e-10)
        assert rec["ch"] == pytest.approx(ch, abs=1e-7)
        assert rec["mnd"] == pytest.approx(orc.mnd(xp), rel=1e-7)
        assert np.abs(opt.x.cpu().numpy() - x).max() <= 1e-6
    assert opt.history[-1]["fx"] < opt.history[0]["fx"]


@pytest.mark.gpu
def test_restart_files_continue_the_run(tmp_path):
    """TopOpt.cc:474-570 / LinearElasticity.cc:447-478: stop after 6 iterations, restart from the files, and the next
    iterations are those of the uninterrupted run; the result container holds every dump of main.cc:114-129"""
    from topopt_in_petsc_amd.driver import TopOpt
    from topopt_in_petsc_amd.mpiio import read_output
    kw = dict(nxyz=(33, 17, 17), nlvls=3, rmin=0.1, volfrac=0.3)
    ref = TopOpt(**kw)
    ref.run(max_itr=8)
    wd = str(tmp_path)
    a = TopOpt(workdir=wd, **kw)
    a.run(max_itr=6)
    out = read_output(os.path.join(wd, "output_00000.dat"))
    assert [d[0] for d in out["dumps"]] == [1, 2, 3, 4, 5, 6, 7]
    assert np.allclose(out["dumps"][-1][2][2], a.xPhys.cpu().numpy(), atol=1e-7)
    assert np.allclose(out["dumps"][-1][1].T.ravel(), a.physics.U.cpu().numpy(), atol=1e-5 * float(a.physics.U.abs().max()))
    b = TopOpt(restartFileVec=os.path.join(wd, "Restart00.dat"), restartFileItr=os.path.join(wd, "Restart00_itr_f0.dat"),
               restartFileVecSol=os.path.join(wd, "RestartSol00.dat"), **kw)
    assert b.itr == 6 and b.fscale == pytest.approx(a.fscale, rel=1e-6)
    b.fscale = a.fscale        # the "%e" companion keeps 7 digits (TopOpt.cc:548); compare the loop itself
    b.run(max_itr=8)
    for r0, r1 in zip(ref.history[6:], b.history):
        assert r0["itr"] == r1["itr"]
        assert r1["fx"] == pytest.approx(r0["fx"], rel=1e-9)
    # like the reference, xold is not part of the restart set (TopOpt.cc:380-381): the first ch is against volfrac
    assert b.history[1]["ch"] == pytest.approx(ref.history[7]["ch"], rel=1e-7, abs=1e-12)
    assert float((b.x - ref.x).abs().max()) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("m", [2, 6])
def test_device_mma_several_constraints(orc, m):
    """m > 1 constraints (MMA.cc:742-880: the dual is an m x m Newton system): the device update against the
    oracle's reference-order sums, 1e-11.  m = 6 also exercises the (m + m^2) x 1024 block-partial buffer."""
    import torch
    import topopt_in_petsc_amd as tp
    c, v = _toy(12 * 8 * 8, seed=3)
    n = c.size
    rng = np.random.default_rng(11)
    W = rng.random((m, n)) + 0.5                      # constraint j: sum(W_j x) / sum(W_j) <= v_j
    W /= W.sum(axis=1, keepdims=True)
    vj = v * (1.0 + 0.05 * np.arange(m))
    grid = tp.Grid(13, 9, 9, 0.125)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    x = np.full(n, v)
    xd = dev(x)
    m_o, m_d = orc.MMA(x, m), tp.MMA(grid, xd, m)
    xmin_d, xmax_d = grid.elem_vec(), grid.elem_vec()
    for it in range(6):
        df = -c / x ** 2 * (10.0 / 2000.0)
        g = [float(W[j] @ x - vj[j]) for j in range(m)]
        dg = [W[j].copy() for j in range(m)]
        xmin, xmax = m_o.SetOuterMovelimit(1e-3, 1.0, 0.2, x)
        m_d.SetOuterMovelimit(1e-3, 1.0, 0.2, xd, xmin_d, xmax_d)
        x = m_o.Update(x, df, g, dg, xmin, xmax)
        m_d.Update(xd, dev(df), g, [dev(d) for d in dg], xmin_d, xmax_d)
        assert m_d.last_inner == m_o.last_inner
        xg = xd.cpu().numpy()
        assert np.abs(xg - x).max() <= 1e-11, (it, np.abs(xg - x).max())
        lam_d, lam_o = np.asarray(m_d.state()[0])[:m], np.asarray(m_o.state()[0])[:m]
        assert np.abs(lam_d - lam_o).max() <= 1e-9 * max(np.abs(lam_o).max(), 1.0)
        xd.copy_(dev(x))


REF_MMA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "host", "_refbuild", "ref_mma")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_MMA), reason="host/_refbuild/ref_mma not built (build container only)")
@pytest.mark.parametrize("m,nproc", [(1, 1), (2, 1), (3, 2), (6, 1)])
def test_device_mma_against_the_references_own_mma_class(tmp_path, m, nproc):
    """The REFERENCE's MMA.cc (compiled unchanged against the compat layer, host/ref_mma_driver.cc; on one or two slab
    processes) and the device MMA on the same synthetic problem with m constraints: the design vector of every
    iteration."""
    import subprocess
    import torch
    import topopt_in_petsc_amd as tp
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ex, ey, ez, iters = 16, 8, 8, 8
    out = str(tmp_path / "x.bin")
    r = subprocess.run([os.path.join(root, "host", "slabrun"), "-n", str(nproc), "--same-device", REF_MMA, str(ex), str(ey), str(ez), str(m),
                        str(iters), out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    raw = open(out, "rb").read()
    n = ex * ey * ez
    xs_ref = [np.frombuffer(raw, dtype=">f8", count=n, offset=k * (8 + 8 * n) + 8).astype(np.float64) for k in range(iters)]
    grid = tp.Grid(ex + 1, ey + 1, ez + 1, 1.0 / ey)
    i = torch.arange(n, dtype=torch.float64, device="cuda")
    x = torch.full((n,), 0.3, dtype=torch.float64, device="cuda")
    xmin, xmax = torch.zeros_like(x), torch.zeros_like(x)
    mma = tp.MMA(grid, x, m)
    a = 1.0 + 0.3 * torch.sin(0.37 * i)
    w = [1.0 + 0.5 * torch.cos(0.11 * i * (j + 1)) for j in range(m)]
    worst = 0.0
    for k in range(iters):
        dfdx = -a / ((x + 0.1) * (x + 0.1))
        dgdx = [(wj / n).contiguous() for wj in w]
        gx = [float((wj * x).sum() / n) - (0.25 + 0.05 * j) for j, wj in enumerate(w)]
        mma.SetOuterMovelimit(0.0, 1.0, 0.2, x, xmin, xmax)
        mma.Update(x, dfdx.contiguous(), gx, dgdx, xmin, xmax)
        worst = max(worst, float(np.abs(x.cpu().numpy() - xs_ref[k]).max()))
    assert worst <= 1e-9, worst

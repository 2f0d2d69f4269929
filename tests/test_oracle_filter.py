"""Filter pins: H symmetric, (H 1)./Hs == 1, forward /
adjoint identity, PDE filter preserves constants and is self-adjoint."""
import numpy as np
import pytest

from tests import scipy_check as sc


@pytest.mark.parametrize("rfac,conn,nnz_int", [(1.5, 1, 19), (2.56, 2, 81), (3.2, 3, 147)])
def test_filter_matrix(orc, rfac, conn, nnz_int):
    ex, ey, ez = 12, 8, 8
    h = 1.0 / ey
    f = orc.Filter(ex + 1, ey + 1, ez + 1, h, rfac * h)
    assert f.conn == conn                      # Filter.cc:326
    H = sc.filter_matrix(ex, ey, ez, h, rfac * h)
    assert f.nnz == H.nnz
    hs = f.hs()
    assert np.allclose(hs, np.asarray(H.sum(axis=1)).ravel(), rtol=1e-13)
    # interior rows carry the full stencil
    assert np.diff(H.indptr).max() == nnz_int
    rng = np.random.default_rng(1)
    x = rng.random(ex * ey * ez)
    xt, xp = f.project(1, x)
    assert np.allclose(xt, (H @ x) / hs, rtol=1e-13)
    assert np.array_equal(xt, xp)
    one, _ = f.project(1, np.ones_like(x))
    assert np.abs(one - 1).max() < 1e-14
    # adjoint identity  <Ht x, y> = <x, Ht^T y>,  Ht = diag(1/Hs) H ; Gradients applies Ht^T
    y = rng.random(x.size)
    g = f.gradient(1, x, xt, y)
    assert xt @ y == pytest.approx(x @ g, rel=1e-13)
    # sensitivity filter (type 0): df <- H(df*x)/Hs/x ; forward is a copy
    xt0, _ = f.project(0, x)
    assert np.array_equal(xt0, x)
    g0 = f.gradient(0, x, xt0, y)
    assert np.allclose(g0, (H @ (y * x)) / hs / x, rtol=1e-13)


def test_default_radius_stencil_sizes(orc):
    # rmin = 0.08 on the 64x32x32 default mesh -> 81 nnz/row
    f = orc.Filter(17, 9, 9, 1.0 / 32, 0.08)
    assert f.conn == 2


def test_heaviside_projection(orc):
    x = np.linspace(0, 1, 11)
    for beta, eta in [(0.1, 0.0), (8.0, 0.5), (48.0, 0.3)]:
        y = orc.heaviside(x, beta, eta)
        assert y[0] == pytest.approx(0.0, abs=1e-15) and y[-1] == pytest.approx(1.0, abs=1e-15)
        assert (np.diff(y) >= 0).all() and y[5] > y[2]
        d = orc.heaviside_chain(x, beta, eta)
        fd = (orc.heaviside(x + 1e-7, beta, eta) - orc.heaviside(x - 1e-7, beta, eta)) / 2e-7
        assert np.allclose(d, fd, rtol=1e-5, atol=1e-8)
    assert orc.mnd(np.array([0.0, 1.0, 0.5, 0.5])) == 0.5
    xt, xp = orc.Filter(9, 5, 5, 0.25, 0.3).project(1, np.full(128, 0.4), proj=True, beta=4.0, eta=0.5)
    assert np.allclose(xp, orc.heaviside(xt, 4.0, 0.5))


def test_pde_filter(orc):
    import scipy.sparse.linalg as spla
    ex, ey, ez = 16, 8, 8
    h = 1.0 / ey
    rmin = 2.56 * h
    pf = orc.PDEFilter(ex + 1, ey + 1, ez + 1, h, rmin, nlv=3)
    rng = np.random.default_rng(2)
    x = rng.random(ex * ey * ez)
    xt, its, hist = pf.apply(x, rtol=1e-12, maxit=200)
    assert 0 < its < 60
    # independent: x~ = T^T K^-1 (vol * T x) with a direct solve
    kf, tf = orc.pde_kf(h, h, h, rmin / 2 / np.sqrt(3))
    K = sc.assemble(ex, ey, ez, kf, dof=1)
    T = sc.elem_to_node_T(ex, ey, ez)
    ref = T.T @ spla.spsolve(K.tocsc(), h ** 3 * (T @ x))
    assert np.abs(xt - ref).max() < 1e-10
    # preserves constants, self-adjoint, smooths, stays within [min, max]
    c, _, _ = pf.apply(np.full(x.size, 0.37), rtol=1e-12, maxit=200)
    assert np.abs(c - 0.37).max() < 1e-10
    y = rng.random(x.size)
    yt, _, _ = pf.apply(y, rtol=1e-12, maxit=200)
    assert xt @ y == pytest.approx(x @ yt, rel=1e-9)
    assert xt.std() < x.std()
    # reference tolerances (rtol 1e-8, <= 60 its, PDEFilter.cc:280-283) converge
    xt2, its2, _ = pf.apply(x)
    assert 0 < its2 <= 60 and np.abs(xt2 - ref).max() < 1e-6

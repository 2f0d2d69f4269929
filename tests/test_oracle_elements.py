"""Pins the oracle's element matrices against the reference's own arithmetic.

tests/golden/ref_ke.bin / ref_kf.bin were produced by running the reference's
Hex8Isoparametric (LinearElasticity.cc:841-998) and PDEFilterMatrix
(PDEFilter.cc:472-576) in the build container (tests/golden/make_ref_vectors.sh).
"""
import os

import numpy as np

G = os.path.join(os.path.dirname(__file__), "golden")


def _ke_cases():
    raw = np.fromfile(os.path.join(G, "ref_ke.bin"))
    return raw.reshape(-1, 4 + 576)


def test_ke_bit_exact_vs_reference(orc):
    cases = _ke_cases()
    assert len(cases) == 6
    for row in cases:
        dx, dy, dz, nu = row[:4]
        ke = orc.hex8_ke_box(dx, dy, dz, nu)
        assert np.array_equal(ke, row[4:]), "KE differs from the reference's bits for %s" % (row[:4],)


def test_ke_known_answers(orc):
    # unit cube, nu = 0.3
    ke = orc.hex8_ke_box(1.0, 1.0, 1.0, 0.3)
    assert ke[0] == 0.23504273504273507
    assert ke[1] == 0.080128205128205107
    assert ke[3] == -0.10683760683760686
    K = ke.reshape(24, 24)
    assert abs(np.trace(K) - 5.6410256410256423) < 1e-14
    assert np.abs(K - K.T).max() < 1e-16          # symmetric to rounding only
    assert np.abs(K.sum(axis=1)).max() < 1e-15    # rigid translation
    # KE scales with h for cubes
    h = 1.0 / 24
    assert np.allclose(orc.hex8_ke_box(h, h, h, 0.3), ke * h, rtol=1e-13, atol=0)
    # 10 distinct magnitudes
    mags = np.unique(np.round(np.abs(K[np.abs(K) > 1e-12]), 10))
    assert len(mags) == 10


def test_ke_rigid_body_modes(orc):
    h = 1.0 / 32
    K = orc.hex8_ke_box(h, h, h, 0.3).reshape(24, 24)
    X = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]]) * h
    # rotations about z, y, x
    for ax in range(3):
        u = np.zeros((8, 3))
        a, b = [(0, 1), (0, 2), (1, 2)][ax]
        u[:, a], u[:, b] = -X[:, b], X[:, a]
        assert np.abs(K @ u.ravel()).max() < 1e-15
    w = np.linalg.eigvalsh(0.5 * (K + K.T))
    assert (w > -1e-15).all() and (w > 1e-10).sum() == 18


def test_kf_bit_exact_vs_reference(orc):
    raw = np.fromfile(os.path.join(G, "ref_kf.bin")).reshape(-1, 4 + 64 + 8)
    assert len(raw) == 12
    for row in raw:
        dx, dy, dz, rmin = row[:4]
        kf, tf = orc.pde_kf(dx, dy, dz, rmin / 2.0 / np.sqrt(3))
        assert np.array_equal(kf, row[4:68])
        assert np.array_equal(tf, row[68:])


def test_kf_known_answers(orc):
    # h = 1/32, rmin = 0.08
    h = 1.0 / 32
    kf, tf = orc.pde_kf(h, h, h, 0.08 / 2.0 / np.sqrt(3))
    assert kf[0] == 6.6858362268518521e-06
    assert kf[1] == 5.6514033564814812e-07
    assert kf[6] == -1.2476038049768519e-06
    K = kf.reshape(8, 8)
    assert np.allclose(K.sum(axis=1), h ** 3 / 8, rtol=1e-13)
    assert np.array_equal(K, K.T)
    assert (tf == 0.125).all()


def test_lambda_bound(orc):
    ke = orc.hex8_ke_box(1.0, 1.0, 1.0, 0.3)
    K = ke.reshape(24, 24)
    d = np.sqrt(np.diag(K))
    ref = np.linalg.eigvalsh(0.5 * (K + K.T) / np.outer(d, d)).max()
    assert abs(orc.elem_lambda_bound(ke) - ref) < 1e-12


def test_effective_element_matrix_restatement():
    """oracle/ke_effective.py (the element matrix the HIP fine-level kernels apply, restated on the host): for the
    reference's KE of a cube and of a box it is symmetric, stays within 1e-15 max|KE| of KE entrywise -- as close to KE as KE is
    to its own formula evaluated in 80-bit arithmetic (oracle/arbiter.py) -- keeps KE's energy on strain modes, and (round 6)
    answers a rigid translation with KE's own MEAN answer: the energy of a translation, sum of the 64 entries of a component
    block (~1e-16 max|KE| of rounding residue in the reference's KE, LinearElasticity.cc:841-998), is reproduced to the rounding of
    one double, spread evenly over the 8 nodes.  The packed form of rounds 1-5 (translation_residue=False) gave exact zeros."""
    import numpy as np
    from oracle import arbiter as arb
    from oracle import oracle as orc
    from oracle.ke_effective import ke_effective
    LD = np.longdouble
    for dims in ((1.0 / 128,) * 3, (0.5, 0.25, 0.125)):
        KE = orc.hex8_ke_box(*dims, 0.3)
        kf = ke_effective(KE)
        assert kf.dtype == np.longdouble and kf.shape == (576,)
        K2, F2 = KE.reshape(24, 24), kf.reshape(24, 24)
        F0 = ke_effective(KE, translation_residue=False).reshape(24, 24)
        mx = np.abs(KE).max()
        assert float(np.abs(F2 - F2.T).max()) == 0.0 and float(np.abs(F0 - F0.T).max()) == 0.0
        assert float(np.abs(F0.sum(axis=1)).max()) == 0.0 and np.abs(K2.sum(axis=1)).max() > 1e-17 * mx
        for c in range(3):
            t = np.zeros(24, dtype=LD)
            t[c::3] = 1.0
            assert float(np.abs(F0 @ t).max()) <= 1e-18 * mx          # rounds 1-5: zero force
            e_ke = K2[c::3, c::3].astype(LD).sum()                     # KE's translation energy (exact in 80 bits)
            assert abs(float(e_ke)) >= 1e-17 * mx                      # ... is rounding residue, not zero
            f = (F2 @ t)[c::3]                                         # the packed form's answer along c: the same at every node
            assert float(np.abs(f - f[0]).max()) <= 1e-19 * mx
            assert abs(float(t @ (F2 @ t) - LD(float(e_ke / 64)) * 64)) <= 1e-19 * mx   # one rounding to double of e / 64
            assert abs(float(t @ (F2 @ t) / e_ke - 1)) <= 2.3e-16
        d_eff = float(np.abs(kf - KE).max()) / mx
        d_ref = float(np.abs(KE - arb.hex8_ke_box(*dims, 0.3)).max()) / mx
        assert 0 < d_eff <= 1e-15 and d_ref <= 1e-15 and d_eff <= 4 * d_ref + 2e-16
        rng = np.random.default_rng(3)
        u = rng.standard_normal(24)
        assert abs(float(u @ (F2 @ u)) / (u @ K2 @ u) - 1) <= 1e-14


def test_krylov_element_matrix_restatement():
    """oracle/ke_effective.py: ke_krylov -- the element matrix of the library's Krylov operator (the plain products; DESIGN 2.1,
    round 6): the packed form plus the translation mode's column and row of T KE T / 64 as the reference's KE
    (LinearElasticity.cc:841-998) has them.  Its answer to a rigid translation (24 x 3) and the translation mode's answer to any
    field (3 x 24) are KE's to the rounding of one double per entry of the transformed matrix (1e-19 max|KE|), where the packed form
    alone is 3e-16 away; entrywise it stays within 1e-15 max|KE| of KE, and it equals the packed form on every pair of non-translation
    modes."""
    import numpy as np
    from oracle import oracle as orc
    from oracle.ke_effective import ke_effective, ke_krylov
    LD = np.longdouble
    for dims in ((1.0 / 128,) * 3, (0.5, 0.25, 0.125)):
        KE = orc.hex8_ke_box(*dims, 0.3)
        K = KE.reshape(24, 24).astype(LD)
        Fe, Fk = ke_effective(KE).reshape(24, 24), ke_krylov(KE).reshape(24, 24)
        mx = float(np.abs(KE).max())
        assert 0 < float(np.abs(Fk - K).max()) <= 1e-15 * mx
        for c in range(3):
            t = np.zeros(24, dtype=LD)
            t[c::3] = 1.0
            assert float(np.abs(Fk @ t - K @ t).max()) <= 1e-18 * mx and float(np.abs(Fe @ t - K @ t).max()) >= 5e-17 * mx     # column
            assert float(np.abs(t @ Fk - t @ K).max()) <= 1e-18 * mx and float(np.abs(t @ Fe - t @ K).max()) >= 5e-17 * mx     # row
        # strain modes (zero mean per component) to strain modes: the packed form
        rng = np.random.default_rng(5)
        u, v = rng.standard_normal(24), rng.standard_normal(24)
        for c in range(3):
            u[c::3] -= u[c::3].mean()
            v[c::3] -= v[c::3].mean()
        assert abs(float(v.astype(LD) @ ((Fk - Fe) @ u.astype(LD)))) <= 1e-17 * mx * np.abs(u).max() * np.abs(v).max() * 24

"""TEST INFRASTRUCTURE (like oracle.py): the element matrix the HIP fine-level kernels APPLY, restated on the host.

The tile kernels (csrc/fine_tile.h, fine_u4.h, matfree_tile.h) evaluate KE u_e in the Walsh-Hadamard basis of the element's
8 nodes: KE_eff = T D T, where D is the block-diagonal part of T KE T / 64 with its symmetric pairs averaged
(csrc/matfree_tile.h: make_sym_ke).  For an exactly box-symmetric KE that is KE itself; the reference's KE
(LinearElasticity.cc:841-998, a 2x2x2 Gauss sum in double) is box symmetric only to rounding -- its rows sum to 7e-16 max|KE|
instead of 0 -- so KE_eff differs from it by 5e-16 max|KE| entrywise, about as much as KE itself differs from the same formula
evaluated in 80-bit arithmetic (5.6e-16).  The compliance of a 128^3 cantilever is sensitive to exactly this kind of
perturbation at the 1e-10 level (the element translation mode has amplitude ~1e3 against strains ~1e-2: any O(eps) change of
how KE answers a rigid translation shows in the 10th digit), so the parity of the HIP path is checked in two steps
(bench.py `parity`, tests/): (1) against the ARBITER run on this KE_eff -- the kernels must reproduce the exact-arithmetic
trajectory of the operator they apply; (2) KE_eff against KE entrywise.

The library exports the same matrix (tp_elasticity_get_ke_effective -> api.LinearElasticity.KE_effective, a double-double
pair); tests/test_gpu_parity.py::test_effective_element_matrix checks the two bit for bit."""
import numpy as np

M2A = [0, 1, 3, 2, 4, 5, 7, 6]


def symke_nz(q, r, s, translation_residue=True):
    if q in (0, 7):
        return True
    if q in (1, 2, 4):
        b = {1: 0, 2: 1, 4: 2}[q]
        # (b, b) of a single-bit class is the element's answer to a rigid translation along b: zero for an exact box
        # element, the mean of the 64 rounding residues of the reference's KE otherwise -- kept since round 6
        return (r != b and s != b) or (translation_residue and r == b and s == b)
    m = {6: 0, 5: 1, 3: 2}[q]
    return (r != m and s != m) or (r == m and s == m)


def _packed_d(KE, translation_residue=True):
    """the packed D (24 x 24, transform order p * 3 + r) as the library's make_sym_ke holds it: doubles"""
    KE = np.asarray(KE, dtype=np.float64).reshape(24, 24)
    pc = lambda v: bin(v).count("1")
    # D = T KE T / 64 in double, term by term as the library's make_sym_ke accumulates it
    D = np.zeros((24, 24))
    for p in range(8):
        for r in range(3):
            for p2 in range(8):
                for s in range(3):
                    acc = 0.0
                    for m in range(8):
                        for m2 in range(8):
                            v = KE[3 * M2A[m] + r, 3 * M2A[m2] + s]
                            acc += -v if (pc(p & m) + pc(p2 & m2)) & 1 else v
                    D[p * 3 + r, p2 * 3 + s] = acc / 64.0
    if translation_residue:
        # the three translation residues are sums of 64 entries that cancel to ~1e-16 of their size: accumulated in 80-bit
        # arithmetic (exact for 64 doubles of one magnitude), then rounded once -- make_sym_ke does the same in `long double`
        for r in range(3):
            acc = np.longdouble(0)
            for m in range(8):
                for m2 in range(8):
                    acc += np.longdouble(KE[3 * M2A[m] + r, 3 * M2A[m2] + r])
            D[r, r] = float(acc / np.longdouble(64))
    Dp = np.zeros((24, 24), dtype=np.longdouble)
    for q in range(8):
        for r in range(3):
            for s in range(r, 3):
                if not symke_nz(q, r, s, translation_residue):
                    continue
                i, j = (q ^ (1 << r)) * 3 + r, (q ^ (1 << s)) * 3 + s
                Dp[i, j] = Dp[j, i] = np.longdouble(0.5 * (D[i, j] + D[j, i]))
    return Dp


def _back_transform(Dp):
    """T^T D T in 80-bit arithmetic, term by term as the library's export sums it (reference dof order)"""
    pc = lambda v: bin(v).count("1")
    out = np.zeros((24, 24), dtype=np.longdouble)
    for m in range(8):
        for r in range(3):
            for m2 in range(8):
                for s in range(3):
                    acc = np.longdouble(0)
                    for p in range(8):
                        for p2 in range(8):
                            v = Dp[p * 3 + r, p2 * 3 + s]
                            acc += -v if (pc(p & m) + pc(p2 & m2)) & 1 else v
                    out[3 * M2A[m] + r, 3 * M2A[m2] + s] = acc
    return out.reshape(-1)


def ke_effective(KE, translation_residue=True):
    """translation_residue=False: the packed form of rounds 1-5 (33 values; rows sum to exactly 0)"""
    return _back_transform(_packed_d(KE, translation_residue))


def ke_krylov(KE):
    """The element matrix of the library's KRYLOV operator (csrc/matfree_tile.h: SYMKE_KRYLOV; library export
    tp_elasticity_get_ke_krylov): ke_effective plus the translation mode's column D[(p,r),(0,s)] and row D[(0,r),(p,s)] of
    D = T KE T / 64, each entry the exact (80-bit) sum of its 64 terms rounded once to double, one-sided (not averaged with its
    mirror image: KE's asymmetry, 1.4e-17, is a tenth of these residues)."""
    KE = np.asarray(KE, dtype=np.float64).reshape(24, 24)
    LD = np.longdouble
    pc = lambda v: bin(v).count("1")
    T = np.zeros((24, 24), dtype=LD)
    for p in range(8):
        for m in range(8):
            for r in range(3):
                T[p * 3 + r, 3 * M2A[m] + r] = -1 if pc(p & m) & 1 else 1
    Dp = _packed_d(KE, True)
    Dx = T @ KE.astype(LD) @ T.T / 64                          # 64-term sums of doubles of one magnitude: exact in 80 bits
    for i in range(24):
        for j in range(24):
            col = j < 3 and not (i < 3 and i == j)
            row = i < 3 and j >= 3
            if col or row:
                Dp[i, j] = LD(float(Dx[i, j]))
    return _back_transform(Dp)

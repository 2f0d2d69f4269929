"""Round 6, CPU only: which entries of the rounding residue of the reference's KE (the part of T KE T / 64 that the
packed Walsh-Hadamard form of the fine kernels drops) carry the 1.6e-10 compliance gap at 128^3?

First-order model: fx = b.U = U^T K U, so a change dK of the operator moves it by -U^T dK U = -sum_e E_e u_e^T dKE u_e
= -trace(dKE G) with the 24x24 moment matrix G = sum_e E_e u_e u_e^T of the converged state -- computed ONCE, in 80-bit
arithmetic; after that every candidate set of restored entries is a 24x24 trace.

usage: r06_ke_residue.py [n] [save.npz]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from oracle.ke_effective import M2A, symke_nz

LD = np.longdouble
pc = lambda v: bin(v).count("1")


def moment_matrix(nx, ny, nz, U, E):
    ex, ey, ez = nx - 1, ny - 1, nz - 1
    Un = U.reshape(nz, ny, nx, 3)
    G = np.zeros((24, 24), dtype=LD)
    # reference corner numbering: 0:(0,0,0) 1:(1,0,0) 2:(1,1,0) 3:(0,1,0), 4..7 one plane up
    off = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
    for k in range(ez):
        ue = np.empty((ey, ex, 24), dtype=LD)
        for a, (dx, dy, dz) in enumerate(off):
            ue[:, :, 3 * a:3 * a + 3] = Un[k + dz, dy:dy + ey, dx:dx + ex, :]
        ue = ue.reshape(-1, 24)
        Ek = E[k * ex * ey:(k + 1) * ex * ey].astype(LD)
        G += (ue * Ek[:, None]).T @ ue
    return G


def wht24():
    """T[(p, r), (m_ref, s)] = delta_rs (-1)^{p.m}; rows in transform order p*3+r, columns in the reference's dof order"""
    T = np.zeros((24, 24), dtype=LD)
    for p in range(8):
        for m in range(8):
            for r in range(3):
                T[p * 3 + r, 3 * M2A[m] + r] = -1 if pc(p & m) & 1 else 1
    return T


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    ex = ey = ez = n
    nlv = 5 if n >= 128 else (4 if n >= 32 else 3)
    cyc = [1, 3, 1, 1][: nlv - 1]
    nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
    t0 = time.time()
    xo = orc.synth_density(ex, ey, ez, h)
    of = orc.Filter(nx, ny, nz, h, 2.56 * h)
    _, xp = of.project(1, xo)
    E = orc.simp(xp)
    KE = orc.hex8_ke_box(h, h, h, 0.3)
    N, R = orc.cantilever_bc(nx, ny, nz, h)
    mg = orc.MG(nx, ny, nz, 3, nlv, 2, 20)
    mg.set_coarse_direct(True)
    mg.set_cycles(cyc)
    mg.assemble(KE, E, N)
    U, its, hist = mg.solve(R * N, rtol=1e-12)
    fx = float((R * N) @ U)
    print("n %d: oracle converged in %d its, fx %.15e (%.0f s)" % (n, its, fx, time.time() - t0), flush=True)
    G = moment_matrix(nx, ny, nz, U, E)
    print("moment matrix done (%.0f s); trace(KE G) / fx - 1 = %.3e" % (time.time() - t0, float((KE.reshape(24, 24).astype(LD) * G).sum() / LD(fx) - 1)))
    if len(sys.argv) > 2:
        np.savez(sys.argv[2], G=G.astype(np.float64), Glo=(G - G.astype(np.float64).astype(LD)).astype(np.float64), KE=KE, fx=fx, n=n)
    analyse(KE, G, fx)


def analyse(KE, G, fx):
    from oracle.ke_effective import ke_effective
    KEl = np.asarray(KE, dtype=np.float64).reshape(24, 24).astype(LD)
    Keff = ke_effective(KE).reshape(24, 24)
    d = Keff - KEl
    print("KE_eff - KE: max %.3e of max|KE|" % float(np.abs(d).max() / np.abs(KEl).max()))
    print("first-order  (fx_eff - fx) / fx = %.4e" % float(-2 * (d * G).sum() / LD(fx)))
    T = wht24()
    # sum d_ij G_ij = sum dh_ij Gh_ij with dh = T d T^T / 64 (the library's D), Gh = T G T^T   (T^T T = 8 I)
    # the objective is evaluated as sum E_e u_e^T KE u_e on the perturbed state: fx moves by -2 U^T dK U
    dh = T @ d @ T.T / 64
    Gh = T @ G @ T.T
    C = -2 * (dh * Gh) / LD(fx)
    print("check: sum of the transformed terms %.4e" % float(C.sum()))
    # by (output mode p, input mode p2)
    print("contribution to (fx_eff - fx)/fx by mode pair (rows: p of the output, columns: p2 of the input), 3x3 blocks summed:")
    B = np.zeros((8, 8))
    for p in range(8):
        for p2 in range(8):
            B[p, p2] = float(C[3 * p:3 * p + 3, 3 * p2:3 * p2 + 3].sum())
    with np.printoptions(precision=2, linewidth=200):
        print(B)
    idx = np.dstack(np.unravel_index(np.argsort(-np.abs(C.astype(np.float64)), axis=None), C.shape))[0]
    print("largest single entries (p r | p2 s : contribution, dh entry / max|D|, Gh entry):")
    Dmax = float(np.abs(T @ KEl @ T.T / 64).max())
    for i, j in idx[:24]:
        print("  (%d %d | %d %d): %+.3e   dh %+.2e  Gh %+.3e" % (i // 3, i % 3, j // 3, j % 3, float(C[i, j]), float(dh[i, j]) / Dmax, float(Gh[i, j])))
    # candidate sets
    def gap_with(restore):
        dd = dh.copy()
        for (i, j) in restore:
            dd[i, j] = 0
        return float(-2 * (dd * Gh).sum() / LD(fx))
    s00 = [(r, s) for r in range(3) for s in range(3)]
    print("restore the (p=0, p2=0) 3x3 block:            gap %.3e" % gap_with(s00))
    row0 = [(r, j) for r in range(3) for j in range(24)] + [(j, r) for r in range(3) for j in range(24)]
    print("restore row and column p = 0 (3x24 twice):    gap %.3e" % gap_with(row0))
    inclass = []
    for i in range(24):
        for j in range(24):
            qi, qj = (i // 3) ^ (1 << (i % 3)), (j // 3) ^ (1 << (j % 3))
            if qi == qj:
                inclass.append((i, j))
    print("restore every in-class entry (8 full 3x3):    gap %.3e" % gap_with(inclass))
    print("restore in-class + the 3x3 block (0, 0):      gap %.3e" % gap_with(inclass + s00))
    print("restore the DIAGONAL of the (0, 0) block only:     gap %.3e" % gap_with([(r, r) for r in range(3)]))
    return dh, Gh, C


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "load":
        z = np.load(sys.argv[2])
        analyse(z["KE"], z["G"].astype(LD) + z["Glo"].astype(LD), float(z["fx"]))
    else:
        main()

"""Round 6, CPU only: which further entries of the residue of T KE T / 64 does the RESIDUAL HISTORY need?  (The compliance is
settled by the three translation residues; at C2 -- 3 levels, 35 iterations -- the late ||r_k|| still move by 1e-9 when KE is
replaced by the packed form.)  The arbiter (80-bit) on KE against the arbiter on KE_eff + a candidate set of restored entries.
usage: r06_arbiter_sets.py ex ey ez nlv [cycles] [direct]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from oracle import arbiter as arb
from oracle.ke_effective import M2A, symke_nz

LD = np.longdouble
pc = lambda v: bin(v).count("1")


def wht24():
    T = np.zeros((24, 24), dtype=LD)
    for p in range(8):
        for m in range(8):
            for r in range(3):
                T[p * 3 + r, 3 * M2A[m] + r] = -1 if pc(p & m) & 1 else 1
    return T


def packed(KE, keep, sym=True):
    """T^T (D restricted to `keep`, symmetrised) T with D = T KE T^T / 64 in 80-bit arithmetic, entries rounded to double"""
    T = wht24()
    D = T @ np.asarray(KE, dtype=np.float64).reshape(24, 24).astype(LD) @ T.T / 64
    Dp = np.zeros((24, 24), dtype=LD)
    for i in range(24):
        for j in range(24):
            if sym and (keep(i, j) or keep(j, i)):
                Dp[i, j] = LD(float(0.5 * (D[i, j] + D[j, i])))
            elif not sym and keep(i, j):
                Dp[i, j] = LD(float(0.5 * (D[i, j] + D[j, i]))) if base(i, j) else LD(float(D[i, j]))
    return (T.T @ Dp @ T).reshape(-1)


def cls(i):
    return (i // 3) ^ (1 << (i % 3))


base = lambda i, j: cls(i) == cls(j) and symke_nz(cls(i), i % 3, j % 3, True)
inclass = lambda i, j: cls(i) == cls(j)
t00 = lambda i, j: i < 3 and j < 3
tcol = lambda i, j: j < 3
lin = lambda i, j: j < 3 and (i // 3) in (1, 2, 4)
if os.environ.get("SETS2"):
    SETS = [
        ("36 + column p in {1,2,4} (one-sided)", lambda i, j: base(i, j) or lin(i, j), False),
        ("36 + column p in {1,2,4} + row", lambda i, j: base(i, j) or lin(i, j), True),
        ("36 + whole column (one-sided)", lambda i, j: base(i, j) or tcol(i, j), False),
    ]
else:
  SETS = [
    ("36 (round 6)", base),
    ("36 + (0 r|0 s) off-diagonal", lambda i, j: base(i, j) or t00(i, j)),
    ("all in-class (8 full 3x3)", lambda i, j: inclass(i, j)),
    ("all in-class + (0|0) block", lambda i, j: inclass(i, j) or t00(i, j)),
    ("36 + translation column and row", lambda i, j: base(i, j) or tcol(i, j)),
]

ex, ey, ez, nlv = [int(v) for v in sys.argv[1:5]]
cyc = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 and sys.argv[5] != "-" else None
direct = len(sys.argv) > 6 and sys.argv[6] == "1"
ncoarse = int(sys.argv[7]) if len(sys.argv) > 7 else 45
nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
xo = orc.synth_density(ex, ey, ez, h)
of = orc.Filter(nx, ny, nz, h, 2.56 * h)
_, xp = of.project(1, xo)
E = orc.simp(xp)
KE = orc.hex8_ke_box(h, h, h, 0.3)
N, R = orc.cantilever_bc(nx, ny, nz, h)
b = (R * N).astype(LD)
mg = arb.MG(nx, ny, nz, 3, nlv, 2, ncoarse)
mg.set_coarse_direct(direct)
if cyc:
    mg.set_cycles(cyc)
t0 = time.time()


def run(kf, kry=None):
    mg.assemble(KE.astype(LD), E.astype(LD), N.astype(LD))
    mg.set_krylov_operator(None)
    if kf is not None:
        mg.reassemble_fine(kf)
    if kry is not None:
        mg.set_krylov_operator(kry)
    U, its, hist = mg.solve(b, rtol=1e-5)
    fx = arb.compliance_sens(nx, ny, nz, KE.astype(LD), U, xp.astype(LD))[0]
    return its, hist, fx


ref = run(None)
print("KE: its %d (%.0f s)" % (ref[0], time.time() - t0), flush=True)
if os.environ.get("SPLIT"):
    # the two uses of the fine-level operator apart: Krylov products (A x0, A p) / the preconditioner's smoother and residual
    k36, kcol = packed(KE, base), packed(KE, lambda i, j: base(i, j) or tcol(i, j), False)
    for tag, kf, kry in (("Krylov: KE, preconditioner: 36", k36, KE.astype(LD)), ("Krylov: 36, preconditioner: KE", None, k36),
                         ("Krylov: 36 + column, preconditioner: 36", k36, kcol), ("Krylov: 36, preconditioner: 36 + column", kcol, k36)):
        a = run(kf, kry)
        k = min(len(a[1]), len(ref[1]))
        e = np.abs(a[1][:k] / ref[1][:k] - 1).astype(np.float64)
        print("%-42s its %d  hist max %.2e (first 10: %.2e, at k = %d)  fx %.2e  (%.0f s)" % (tag, a[0], e.max(), e[:10].max(), int(e.argmax()), float(abs(a[2] / ref[2] - 1)), time.time() - t0), flush=True)
    sys.exit(0)
if os.environ.get("SPLIT2"):
    T = wht24()
    D = T @ KE.reshape(24, 24).astype(LD) @ T.T / 64
    def pk(keep, avg):
        Dp = np.zeros((24, 24), dtype=LD)
        for i in range(24):
            for j in range(24):
                if keep(i, j):
                    Dp[i, j] = LD(float(0.5 * (D[i, j] + D[j, i]))) if (avg and base(i, j)) else LD(float(D[i, j]))
        return (T.T @ Dp @ T).reshape(-1)
    k36 = packed(KE, base)
    trow = lambda i, j: i < 3
    for tag, kry in (("all 576 entries of D, rounded to double", pk(lambda i, j: True, False)),
                     ("36 unaveraged + column", pk(lambda i, j: base(i, j) or tcol(i, j), False)),
                     ("36 averaged + column + row", pk(lambda i, j: base(i, j) or tcol(i, j) or trow(i, j), True)),
                     ("all in-class unaveraged + column + row", pk(lambda i, j: inclass(i, j) or tcol(i, j) or trow(i, j), False)),
                     ("36 + column + row + linear-mode block (p, p2 in 0,1,2,4)", pk(lambda i, j: base(i, j) or ((i // 3) in (0, 1, 2, 4) and (j // 3) in (0, 1, 2, 4)), True))):
        a = run(k36, kry)
        k = min(len(a[1]), len(ref[1]))
        e = np.abs(a[1][:k] / ref[1][:k] - 1).astype(np.float64)
        print("Krylov: %-58s its %d  hist max %.2e (first 10: %.2e, at k = %d)  fx %.2e  (%.0f s)" % (tag, a[0], e.max(), e[:10].max(), int(e.argmax()), float(abs(a[2] / ref[2] - 1)), time.time() - t0), flush=True)
    sys.exit(0)
for item in SETS:
    tag, keep = item[0], item[1]
    sym = item[2] if len(item) > 2 else True
    n = sum(1 for i in range(24) for j in range(i, 24) if keep(i, j) or keep(j, i))
    a = run(packed(KE, keep, sym))
    k = min(len(a[1]), len(ref[1]))
    e = np.abs(a[1][:k] / ref[1][:k] - 1).astype(np.float64)
    print("%-36s %3d values: its %d  hist max %.2e (first 10: %.2e, at k = %d)  fx %.2e  (%.0f s)" % (tag, n, a[0], e.max(), e[:10].max(), int(e.argmax()), float(abs(a[2] / ref[2] - 1)), time.time() - t0), flush=True)

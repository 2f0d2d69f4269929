"""Round 6, CPU only: the arbiter (80-bit) on KE, on the packed form of rounds 1-5 (33 values) and on the packed form with the
three translation residues restored (36 values): residual history and compliance, rtol 1e-5 and 1e-12.
usage: r06_arbiter_ke.py [n]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from oracle import arbiter as arb
from oracle.ke_effective import ke_effective

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ex = ey = ez = n
nlv = 5 if n >= 128 else (4 if n >= 32 else 3)
cyc = [1, 3, 1, 1][: nlv - 1]
nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
LD = np.longdouble
xo = orc.synth_density(ex, ey, ez, h)
of = orc.Filter(nx, ny, nz, h, 2.56 * h)
_, xp = of.project(1, xo)
E = orc.simp(xp)
KE = orc.hex8_ke_box(h, h, h, 0.3)
N, R = orc.cantilever_bc(nx, ny, nz, h)
b = (R * N).astype(LD)
mg = arb.MG(nx, ny, nz, 3, nlv, 2, 20)
mg.set_coarse_direct(True)
mg.set_cycles(cyc)
res = {}
t0 = time.time()
for tag, kf in (("KE", None), ("packed33", ke_effective(KE, False)), ("packed36", ke_effective(KE, True))):
    mg.assemble(KE.astype(LD), E.astype(LD), N.astype(LD))
    if kf is not None:
        mg.reassemble_fine(kf)
    out = {}
    for rt in (1e-5, 1e-12):
        U, its, hist = mg.solve(b, rtol=rt)
        fx = arb.compliance_sens(nx, ny, nz, KE.astype(LD), U, xp.astype(LD))[0]
        out[rt] = (its, hist, fx)
    res[tag] = out
    print("%-9s its %d / %d  fx %.18Lg  (%.0f s)" % (tag, out[1e-5][0], out[1e-12][0], out[1e-5][2], time.time() - t0), flush=True)
for tag in ("packed33", "packed36"):
    for rt in (1e-5, 1e-12):
        a, r = res[tag][rt], res["KE"][rt]
        k = min(len(a[1]), len(r[1]))
        print("%s vs KE, rtol %g: its %d/%d  hist max rel %.3e  fx rel %.3e" % (tag, rt, a[0], r[0], float(np.abs(a[1][:k] / r[1][:k] - 1).max()), float(abs(a[2] / r[2] - 1))))

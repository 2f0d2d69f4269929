"""Round 6: how far is the residual history of C3 (256x128x128, 6 levels, 1,3,1,1,1) from the oracle's for the variants of the
stencil kernel -- node form with / without mirrored reads, the row form of rounds 1-5?  One process per variant (the switches are
read once), the oracle once.  usage: r06_c3_hist_variants.py [ex ey ez nlv]"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = [int(v) for v in sys.argv[1:5]] if len(sys.argv) > 4 else [256, 128, 128, 6]
ex, ey, ez, nlv = args
cyc = [1, 3, 1, 1, 1, 1][: nlv - 1]
WORKER = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import topopt_in_petsc_amd as tp
ex, ey, ez, nlv = %d, %d, %d, %d
grid = tp.Grid(ex + 1, ey + 1, ez + 1, 1.0 / ey)
le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=nlv, nsmooth=2, ncoarse=20, rtol=1e-5, coarse_direct=1))
le.set_cycles(%r)
le.SetUpLoadAndBC()
flt = tp.Filter(grid, 1, 2.56 / ey)
x = grid.synth_density(12345)
xt, xp, df, dg = grid.elem_vec(), grid.elem_vec(), grid.elem_vec(), grid.elem_vec()
flt.FilterProject(x, xt, xp)
fx, gx = le.ComputeObjectiveConstraintsSensitivities(df, dg, xp, 1e-9, 1.0, 3.0, 0.12, hist_cap=64)
np.savez(sys.argv[1], hist=np.asarray(le.last_hist), fx=fx, its=le.last_its)
""" % (ROOT, ex, ey, ez, nlv, cyc)
d = tempfile.mkdtemp()
res = {}
for tag, env in (("node + mirrored (default)", {}), ("node, own rows only", {"TP_NO_DIA_SYM": "1"}), ("row form (rounds 1-5)", {"TP_DIA_NODE": "0"})):
    e = dict(os.environ)
    for k in ("TP_NO_DIA_SYM", "TP_DIA_NODE"):
        e.pop(k, None)
    e.update(env)
    fn = os.path.join(d, "v%d.npz" % len(res))
    subprocess.run([sys.executable, "-c", WORKER, fn], env=e, check=True, stderr=subprocess.DEVNULL)
    res[tag] = np.load(fn)
from oracle import oracle as orc
nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
xo = orc.synth_density(ex, ey, ez, h)
_, xpo = orc.Filter(nx, ny, nz, h, 2.56 * h).project(1, xo)
KE = orc.hex8_ke_box(h, h, h, 0.3)
N, R = orc.cantilever_bc(nx, ny, nz, h)
mg = orc.MG(nx, ny, nz, 3, nlv, 2, 20)
mg.set_coarse_direct(True)
mg.set_cycles(cyc)
mg.assemble(KE, orc.simp(xpo), N)
U, its, hist = mg.solve(R * N, rtol=1e-5)
for tag, r in res.items():
    k = min(len(hist), len(r["hist"]))
    e = np.abs(r["hist"][:k] / hist[:k] - 1)
    print("%-28s its %d / %d  hist vs oracle: max %.2e (first 10: %.2e, at k = %d)" % (tag, int(r["its"]), its, e.max(), e[:10].max(), int(e.argmax())))
tags = list(res)
for a in range(len(tags)):
    for b in range(a + 1, len(tags)):
        k = min(len(res[tags[a]]["hist"]), len(res[tags[b]]["hist"]))
        print("   %s vs %s: %.2e" % (tags[a], tags[b], np.abs(res[tags[a]]["hist"][:k] / res[tags[b]]["hist"][:k] - 1).max()))

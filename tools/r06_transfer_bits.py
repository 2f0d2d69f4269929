"""One-off check (round 6): the wide-load restriction and the branch-free prolongation against the forms of rounds 1-5 (a library
built with -DTP_RESTRICT_NARROW, loaded through TP_LIB in a second process): restriction, prolongation, one
V-cycle, a whole solve -- the same bits.  usage: r06_transfer_bits.py old_lib.so"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import topopt_in_petsc_amd as tp
tp.load_library()
res = {}
for (ex, ey, ez, nlv) in ((96, 64, 32, 4), (40, 24, 56, 3)):
    grid = tp.Grid(ex + 1, ey + 1, ez + 1, 1.0 / ey)
    le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=nlv, nsmooth=2, ncoarse=20, rtol=1e-8))
    le.SetUpLoadAndBC()
    le.AssembleStiffnessMatrix(grid.synth_density(12345), 1e-9, 1.0, 3.0)
    rng = np.random.default_rng(3)
    tag = "%%dx%%dx%%d_" %% (ex, ey, ez)
    for l in range(nlv - 1):
        rf = torch.from_numpy(rng.standard_normal(3 * le.level_nodes(l))).cuda()
        res[tag + "restrict%%d" %% l] = le.restrict(l, rf).cpu().numpy()
        xc = torch.from_numpy(rng.standard_normal(3 * le.level_nodes(l + 1))).cuda()
        res[tag + "prolong%%d" %% l] = le.prolong_add(l, xc, rf.clone()).cpu().numpy()
    r = torch.from_numpy(rng.standard_normal(3 * le.level_nodes(0))).cuda()
    res[tag + "pc"] = le.precond(r).cpu().numpy()
    its = le.KSPSolve(hist_cap=300)
    res[tag + "U"], res[tag + "hist"] = le.U.cpu().numpy(), np.asarray(le.last_hist)
    grid.close()
np.savez(sys.argv[1], **res)
""" % ROOT
d = tempfile.mkdtemp()
out = {}
for tag, env in (("new", {}), ("old", {"TP_LIB": os.path.abspath(sys.argv[1])})):
    e = dict(os.environ)
    e.pop("TP_LIB", None)
    e.update(env)
    fn = os.path.join(d, tag + ".npz")
    subprocess.run([sys.executable, "-c", WORKER, fn], env=e, check=True)
    out[tag] = np.load(fn)
bad = [k for k in out["new"].files if not np.array_equal(out["new"][k], out["old"][k])]
print("transfer kernels, %d arrays compared bitwise: %s" % (len(out["new"].files), "ALL EQUAL" if not bad else "DIFFERENT: %s" % bad))
sys.exit(1 if bad else 0)

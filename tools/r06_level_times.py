"""Back-to-back time per Chebyshev step of every level of a workload's hierarchy, with the bytes its stored stencil streams.
usage: r06_level_times.py ex ey ez nlv"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import topopt_in_petsc_amd as tp

ex, ey, ez, nlv = [int(v) for v in sys.argv[1:5]]
grid = tp.Grid(ex + 1, ey + 1, ez + 1, 1.0 / ey)
le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=nlv, nsmooth=2, ncoarse=20, coarse_direct=1))
le.SetUpLoadAndBC()
le.AssembleStiffnessMatrix(grid.synth_density(12345), 1e-9, 1.0, 3.0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for l in range(nlv):
    n = 3 * le.level_nodes(l)
    b, x = torch.randn(n, dtype=torch.float64, device="cuda"), torch.zeros(n, dtype=torch.float64, device="cuda")
    def t(k, reps=20):
        le.smooth(l, b, x, k, False)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            le.smooth(l, b, x, k, False)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    per = (t(8) - t(0)) / 8
    st = 249.0 * 8 * n / 3
    print("level %d: %8d nodes  %.1f us per Chebyshev step%s" % (l, n // 3, 1e3 * per, "  (stored stencil: %.0f MB -> %.2f TB/s)" % (st / 1e6, st / per / 1e9) if l >= 2 else ""))

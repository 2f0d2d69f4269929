"""Back-to-back times of the fine level's fused Chebyshev step by loop variant: first step of a sweep (no previous
iterate read: PREV = false) against the later steps (PREV = true), and the residual / apply forms.  128^3 by default."""
import sys, time
import torch
sys.path.insert(0, ".")
import topopt_in_petsc_amd as tp
ex = int(sys.argv[1]) if len(sys.argv) > 1 else 128
grid = tp.Grid(ex + 1, ex + 1, ex + 1, 1.0 / ex)
le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=1))
le.SetUpLoadAndBC()
le.AssembleStiffnessMatrix(grid.synth_density(12345), 1e-9, 1.0, 3.0)
u = grid.node_vec(3).normal_()
y = torch.zeros_like(u)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, n=40):
    t0 = time.time()
    while time.time() - t0 < 0.2:
        for _ in range(5): fn()
        torch.cuda.synchronize()
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
t0 = timed(lambda: le.smooth(0, u, y, 0, False))
t1 = timed(lambda: le.smooth(0, u, y, 1, False))
t2 = timed(lambda: le.smooth(0, u, y, 2, False))
t8 = timed(lambda: le.smooth(0, u, y, 8, False))
tz2 = timed(lambda: le.smooth(0, u, y, 2, True))
ta = timed(lambda: le.MatMult(u, y))
print("copy-only %.1f us | k=1 (PREV=false) %.1f | k=2: second step %.1f | k=8: avg of steps 2..8 %.1f | zero guess k=2 total %.1f | apply %.1f" % (
    t0, t1 - t0, t2 - t1, (t8 - t1) / 7, tz2, ta))

#!/usr/bin/env python
"""Time of the cone filter (forward + one gradient) at a given radius: usage filter_time.py ex ey ez rmin"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import topopt_in_petsc_amd as tp
ex, ey, ez = [int(v) for v in sys.argv[1:4]]
rmin = float(sys.argv[4])
grid = tp.Grid(ex + 1, ey + 1, ez + 1, 1.0 / ey)
f = tp.Filter(grid, 1, rmin)
x = grid.synth_density(12345)
xt, xp, df = grid.elem_vec(), grid.elem_vec(), grid.elem_vec(1.0)
for _ in range(3):
    f.FilterProject(x, xt, xp)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    f.FilterProject(x, xt, xp)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(n):
    f.Gradients(x, xt, df, [])
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%dx%dx%d rmin %g ElemConn %d (%d taps) env %s: project %.1f us  gradient %.1f us" % (ex, ey, ez, rmin, f.ElemConn, (2 * f.ElemConn + 1) ** 3,
      {k: v for k, v in os.environ.items() if k.startswith("TP_")}, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t1) / n))

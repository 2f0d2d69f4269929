#!/usr/bin/env python
"""A/B timing of the fine-level kernels (plain SpMV and fused Chebyshev step) with HIP events.
usage: fine_ab.py ex ey ez [reps]   (variants through the environment: TP_FINE_V, TP_TILE_KZ, ...)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import topopt_in_petsc_amd as tp

ex, ey, ez = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (128, 128, 128)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
h = 1.0 / ey
grid = tp.Grid(ex + 1, ey + 1, ez + 1, h)
le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=1))
le.SetUpLoadAndBC()
x = grid.synth_density()
le.AssembleStiffnessMatrix(x, 1e-9, 1.0, 3.0)
u = grid.node_vec(3).normal_()
y = torch.zeros_like(u)
b = grid.node_vec(3).normal_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, n):
    import time
    t0 = time.time()
    while time.time() - t0 < 0.15:  # clocks settled
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


spmv = timed(lambda: le.MatMult(u, y), reps)
k = 8
t_s = timed(lambda: le.smooth(0, b, y, k, False), max(reps // 4, 2))
t_c = timed(lambda: le.smooth(0, b, y, 0, False), max(reps // 4, 2))
cheb = (t_s - t_c) / k
nn, ne = (ex + 1) * (ey + 1) * (ez + 1), ex * ey * ez
sb, cb = 48.0 * nn + 8.0 * ne, 96.0 * nn + 8.0 * ne
print(json.dumps({"mesh": "%dx%dx%d" % (ex, ey, ez), "env": {k: v for k, v in os.environ.items() if k.startswith("TP_")},
                  "spmv_us": 1e3 * spmv, "spmv_frac": sb / spmv / 1e6 / 8000, "cheb_us": 1e3 * cheb,
                  "cheb_frac": cb / cheb / 1e6 / 8000}))

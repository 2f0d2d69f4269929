#!/usr/bin/env python
"""Micro-benchmark of the fine-level matrix-free SpMV (for rocprofv3 runs).
usage: spmv_bench.py [ex ey ez] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import topopt_in_petsc_amd as tp

ex, ey, ez = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (128, 128, 128)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
h = 1.0 / ey
grid = tp.Grid(ex + 1, ey + 1, ez + 1, h)
le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=1))
le.SetUpLoadAndBC()
x = grid.synth_density()
le.AssembleStiffnessMatrix(x, 1e-9, 1.0, 3.0)
u = grid.node_vec(3).normal_()
y = torch.zeros_like(u)
for _ in range(3):
    le.MatMult(u, y)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(reps):
    le.MatMult(u, y)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
nb = 48.0 * (ex + 1) * (ey + 1) * (ez + 1) + 8.0 * ex * ey * ez
print("spmv %dx%dx%d: %.4f ms  %.1f GB/s algorithmic (%.3f of 8 TB/s)  %.2f TF/s (dense-equivalent flops)" %
      (ex, ey, ez, ms, nb / ms / 1e6, nb / ms / 1e6 / 8000, 1152.0 * ex * ey * ez / ms / 1e9))

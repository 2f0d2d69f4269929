#!/usr/bin/env python
"""Workload for the HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE are collected in
separate rocprofv3 --pmc runs of this script):
  * k_scale on a 1 GiB vector  -> calibration of the counters for 8-byte-per-lane access
    (known traffic: n*8 B read + n*8 B written);
  * the fine-level SpMV at the requested size."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import topopt_in_petsc_amd as tp

ex, ey, ez = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (256, 256, 256)
h = 1.0 / ey
grid = tp.Grid(ex + 1, ey + 1, ez + 1, h)
le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=1))
le.SetUpLoadAndBC()
x = grid.synth_density()
le.AssembleStiffnessMatrix(x, 1e-9, 1.0, 3.0)
n = 1 << 27
v = torch.ones(n, dtype=torch.float64, device="cuda")
for _ in range(3):
    grid.L.tp_vec_scale(grid.handle, v.data_ptr(), 1.0, n)
u = grid.node_vec(3).normal_()
y = torch.zeros_like(u)
for _ in range(5):
    le.MatMult(u, y)
b = grid.node_vec(3).normal_()
le.smooth(0, b, y, 5, False)   # 5 launches of the fused Chebyshev kernel k_matfree_tile<2,0>
torch.cuda.synchronize()
print("calibration bytes read=%d written=%d ; spmv algorithmic bytes=%d" %
      (8 * n, 8 * n, 48 * (ex + 1) * (ey + 1) * (ez + 1) + 8 * ex * ey * ez))

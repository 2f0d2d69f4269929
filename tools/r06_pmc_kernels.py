#!/usr/bin/env python
"""Workload of the round-6 counter passes for the kernels nobody had looked at (VERDICT r5 "next" 8), at the metric's 128^3
mesh with the bench's hierarchy: the cone filter (k_conv_filter_tiled<2>), the 0 <-> 1 transfers (k_restrict<3>,
k_prolong_add<3>), the level-2 block stencil (k_dia_row_split<3,...>), the Krylov product (k_fine_tile<3>) beside the packed
product (k_fine_tile<0>), and k_scale on 1 GiB as the calibration of the byte counters.
usage (one counter set per rocprofv3 pass): rocprofv3 --pmc ... --kernel-trace --output-format csv -d DIR -- python tools/r06_pmc_kernels.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import topopt_in_petsc_amd as tp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
h = 1.0 / n
grid = tp.Grid(n + 1, n + 1, n + 1, h)
le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=5, nsmooth=2, ncoarse=20, coarse_direct=1))
le.set_cycles([1, 3, 1, 1])
le.SetUpLoadAndBC()
flt = tp.Filter(grid, 1, 2.56 * h)
x = grid.synth_density(12345)
xt, xp = grid.elem_vec(), grid.elem_vec()
flt.FilterProject(x, xt, xp)
le.AssembleStiffnessMatrix(xp, 1e-9, 1.0, 3.0)
nn = 1 << 27
v = torch.ones(nn, dtype=torch.float64, device="cuda")
for _ in range(3):
    grid.L.tp_vec_scale(grid.handle, v.data_ptr(), 1.0, nn)
REP = 5
xe, ye = grid.elem_vec().uniform_(), grid.elem_vec()
for _ in range(REP):
    flt.MultH(xe, ye)
u = grid.node_vec(3).normal_()
y = torch.zeros_like(u)
for _ in range(REP):
    le.MatMult(u, y)
    le.MatMultKrylov(u, y)
rf = le.level_vec(0).normal_()
for _ in range(REP):
    rc = le.restrict(0, rf)
xc = le.level_vec(1).normal_()
for _ in range(REP):
    le.prolong_add(0, xc, rf)
b2, x2 = le.level_vec(2).normal_(), le.level_vec(2)
le.smooth(2, b2, x2, REP, False)
b1, x1 = le.level_vec(1).normal_(), le.level_vec(1)
le.smooth(1, b1, x1, REP, False)
torch.cuda.synchronize()
nd0, nd1, nd2, nel = (n + 1) ** 3, (n // 2 + 1) ** 3, (n // 4 + 1) ** 3, n ** 3
print("algorithmic bytes: conv_filter %d  restrict/prolong %d  level2 %d  spmv %d" % (16 * nel, 24 * (nd0 + nd1), 249 * 8 * nd2, 48 * nd0 + 8 * nel))

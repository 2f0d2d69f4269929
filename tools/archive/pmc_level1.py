#!/usr/bin/env python
"""Workload for counter passes over the LEVEL-1 operator (k_matfree_tile<EPI, 1>) at 128^3: 8 fused Chebyshev steps and
5 plain applies on level 1 of a 2-level hierarchy.  usage: pmc_level1.py [ex ey ez]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import topopt_in_petsc_amd as tp

ex, ey, ez = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (128, 128, 128)
grid = tp.Grid(ex + 1, ey + 1, ez + 1, 1.0 / ey)
le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=2))
le.SetUpLoadAndBC()
le.AssembleStiffnessMatrix(grid.synth_density(), 1e-9, 1.0, 3.0)
n1 = 3 * le.level_nodes(1)
b = torch.randn(n1, dtype=torch.float64, device="cuda")
x = torch.randn_like(b)
le.smooth(1, b, x, 8, False)
for _ in range(5):
    le.level_apply(1, x)
torch.cuda.synchronize()

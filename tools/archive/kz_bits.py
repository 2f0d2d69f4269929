"""does the fine kernel's result depend (bitwise) on the z-chunk length?  usage: kz_bits.py out.pt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, topopt_in_petsc_amd as tp
ex, ey, ez = 16, 8, 8
g = tp.Grid(ex + 1, ey + 1, ez + 1, 1.0 / ey)
le = tp.LinearElasticity(g, tp.SolverOptions(nlvls=1)); le.SetUpLoadAndBC()
le.AssembleStiffnessMatrix(g.synth_density(), 1e-9, 1.0, 3.0)
torch.manual_seed(3)
b = torch.randn(3 * 17 * 9 * 9, dtype=torch.float64, device="cuda"); x = torch.randn_like(b)
outs = []
for k, zg in ((1, False), (2, False), (3, False), (4, True)):
    a = x.clone(); le.smooth(0, b, a, k, zg); outs.append(a.cpu())
outs.append(le.MatMult(x).cpu())
torch.save(outs, sys.argv[1])

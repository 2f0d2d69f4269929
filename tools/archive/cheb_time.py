import os, sys
sys.path.insert(0, '/root/repo')
import torch, topopt_in_petsc_amd as tp
ex=ey=ez=128; h=1.0/ey
grid=tp.Grid(ex+1,ey+1,ez+1,h); le=tp.LinearElasticity(grid,tp.SolverOptions(nlvls=1, ncoarse=9))
le.SetUpLoadAndBC(); x=grid.synth_density(); le.AssembleStiffnessMatrix(x,1e-9,1.0,3.0)
r=grid.node_vec(3).normal_()
for _ in range(2): le.precond(r)
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(5): le.precond(r)
e1.record(); torch.cuda.synchronize()
print("precond (1 first + 8 fused Chebyshev steps on the fine level): %.1f us per Chebyshev step" % (e0.elapsed_time(e1)/5/8*1e3))

import sys; sys.path.insert(0, '/root/repo')
import topopt_in_petsc_amd as tp
for (ex,ey,ez,nlv) in ((48,24,24,4),(128,128,128,4)):
    h=1.0/ey; g=tp.Grid(ex+1,ey+1,ez+1,h); le=tp.LinearElasticity(g,tp.SolverOptions(nlvls=nlv)); le.SetUpLoadAndBC()
    x=g.elem_vec(0.12) if ex==48 else g.synth_density(12345)
    le.AssembleStiffnessMatrix(x,1e-9,1.0,3.0)
    its=le.KSPSolve(hist_cap=64)
    print("MESH %dx%dx%d its %d" % (ex,ey,ez,its)); print(le.petsc_options()); print("hist", " ".join("%.6e" % v for v in le.last_hist[:its+1]))

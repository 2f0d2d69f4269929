import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from oracle import oracle as orc
ex = ey = ez = int(sys.argv[1]); nlv = int(sys.argv[2]); ns = int(sys.argv[3]); nc = int(sys.argv[4])
nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
x = orc.synth_density(ex, ey, ez, h)
flt = orc.Filter(nx, ny, nz, h, 2.56 * h)
xt, xp = flt.project(1, x)
KE = orc.hex8_ke_box(h, h, h, 0.3)
N, R = orc.cantilever_bc(nx, ny, nz, h)
E = orc.simp(xp)
for lo, hi in [(0.1, 1.1), (0.05, 1.1), (0.2, 1.1), (0.3, 1.1), (0.1, 1.05), (0.2, 1.05), (0.15, 1.1), (0.1, 1.2)]:
    t0 = time.time()
    mg = orc.MG(nx, ny, nz, 3, nlv, ns, nc, lo, hi)
    mg.assemble(KE, E, N)
    U, its, hist = mg.solve(R * N, rtol=1e-5)
    print("lo %.2f hi %.2f : its %d  (%.1f s)" % (lo, hi, its, time.time() - t0), flush=True)

"""Time of one cone-filter application (FilterProject) at the reference's absolute default radius rmin = 0.08 on the meshes
where it exceeds ElemConn 8: z-streamed kernel against the direct loop (TP_NO_FILTER_TILE=1 in another process)."""
import sys, time
import torch
sys.path.insert(0, ".")
import topopt_in_petsc_amd as tp
for (ex, ey, ez) in ((128, 128, 128), (256, 128, 128)):
    grid = tp.Grid(ex + 1, ey + 1, ez + 1, 1.0 / ey)
    f = tp.Filter(grid, 1, 0.08)
    x = grid.synth_density(12345)
    xt, xp = grid.elem_vec(), grid.elem_vec()
    f.FilterProject(x, xt, xp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        f.FilterProject(x, xt, xp)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 3
    taps = (2 * f.ElemConn + 1) ** 3
    print("%dx%dx%d rmin 0.08: ElemConn %d (%d taps), FilterProject %.3f ms = %.1f Tfma/s" % (ex, ey, ez, f.ElemConn, taps, ms, ex * ey * ez * taps / ms / 1e9))
    f.close(); grid.close()

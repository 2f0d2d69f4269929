import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import topopt_in_petsc_amd as tp
for (ex, ey, ez, rf) in ((128, 128, 128, 2.56), (128, 128, 128, 1.5), (128, 128, 128, 3.5), (128, 64, 64, 2.56), (48, 24, 24, 2.56), (50, 27, 21, 2.56)):
    h = 1.0 / ey
    grid = tp.Grid(ex + 1, ey + 1, ez + 1, h)
    flt = tp.Filter(grid, 1, rf * h)
    x = grid.synth_density(12345); y = grid.elem_vec()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20): flt.MultH(x, y)
    torch.cuda.synchronize(); e0.record()
    for _ in range(200): flt.MultH(x, y)
    e1.record(); torch.cuda.synchronize()
    print("%dx%dx%d ElemConn %d: %.1f us  checksum %.17g" % (ex, ey, ez, flt.ElemConn, 1e3 * e0.elapsed_time(e1) / 200, float(y.double().sum())))
    np.save("/tmp/filt_%d_%d_%d_%s.npy" % (ex, ez, flt.ElemConn, os.environ.get("TP_FILTER_ZMULTI", "d")), y.cpu().numpy())
    grid.close()

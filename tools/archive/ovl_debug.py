"""debug: overlapped vs blocking halos, op by op (2+ ranks on one GPU, gloo).  torchrun --nproc-per-node 2 tools/ovl_debug.py"""
import os, sys
import numpy as np, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo"); rank, world = dist.get_rank(), dist.get_world_size()
import topopt_in_petsc_amd as tp
torch.cuda.set_device(0)
ex, ey, ez, nlv = 16, 8, 16, 3
nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
def mk(ovl):
    if not ovl: os.environ["TP_OVERLAP"] = "0"
    g = tp.Grid(nx, ny, nz, h, rank=rank, nranks=world)
    os.environ.pop("TP_OVERLAP", None)
    le = tp.LinearElasticity(g, tp.SolverOptions(nlvls=nlv, rtol=1e-9, max_it=300)); le.SetUpLoadAndBC()
    le.AssembleStiffnessMatrix(g.synth_density(), 1e-9, 1.0, 3.0)
    return g, le
g1, l1 = mk(True); g0, l0 = mk(False)
torch.manual_seed(5 + rank)
for lev in range(nlv):
    n = 3 * l1.level_nodes(lev)
    b = torch.randn(n, dtype=torch.float64, device="cuda"); x = torch.randn(n, dtype=torch.float64, device="cuda")
    for k, zg in ((1, False), (2, False), (3, False), (4, True)):
        a = x.clone(); c = x.clone()
        l1.smooth(lev, b, a, k, zg); l0.smooth(lev, b, c, k, zg)
        own = g1.part.level(lev).owned_slice(3)
        d = (a[own] - c[own]).abs().max().item()
        if d > 0:
            pl = 3 * g1.part.level(lev).plane
            dif = (a - c).abs()
            bad = torch.nonzero(dif > 0).flatten()
            planes = torch.unique(bad // pl).tolist()
            print("rank %d level %d k=%d differing planes %s (own_lo %d own_hi %d) n=%d" % (rank, lev, k, planes[:12], g1.part.level(lev).own_lo, g1.part.level(lev).own_hi, bad.numel()), flush=True)
        print("rank %d level %d k=%d zero=%s maxdiff %.3e overlapped=%d" % (rank, lev, k, zg, d, g1.halo_overlap), flush=True)
r = torch.randn(3 * l1.level_nodes(0), dtype=torch.float64, device="cuda")
z1, z0 = l1.precond(r), l0.precond(r)
own = g1.part.owned_slice(3)
print("rank %d precond maxdiff %.3e" % (rank, (z1[own] - z0[own]).abs().max().item()), flush=True)
dist.destroy_process_group()

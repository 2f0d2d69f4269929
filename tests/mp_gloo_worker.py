"""world_size-2 worker (launched by torch.distributed.run from the tests).

mode cpu : no GPU.  Checks the z-slab partition + SlabComm halo/all-reduce logic by
           running a slab-decomposed matrix-free CG built from the ORACLE's element
           kernel and comparing with the serial oracle solve.
mode gpu : both ranks share cuda:0, gloo backend with host staging.  Runs the real
           HIP solver on 2 slabs and compares with the oracle (residual history,
           U, objective, sensitivities, filters).
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from topopt_in_petsc_amd.partition import SlabPartition  # noqa: E402
from topopt_in_petsc_amd.comm import SlabComm  # noqa: E402


def cpu_mode(rank, world):
    ex, ey, ez = 12, 6, 8
    nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
    part = SlabPartition(nx, ny, nz, rank, world)
    comm = SlabComm(part, "cpu")
    KE = orc.hex8_ke_box(h, h, h, 0.3)
    N, R = orc.cantilever_bc(nx, ny, nz, h)
    x = orc.synth_density(ex, ey, ez, h)
    E = orc.simp(x)
    b = R * N
    # ---- local slab data (own + ghost planes; own element layers + one ghost layer above)
    gs = part.global_slice(3)
    Nl, bl = N[gs].copy(), b[gs].copy()
    lay = ex * ey
    ezl = part.ez_own + (1 if part.has_hi else 0)
    El = E[lay * part.elem_z0: lay * (part.elem_z0 + ezl)].copy()
    own = part.owned_slice(3)

    def apply_local(u):
        """owned rows of (N K N + I - N) u from local data: ghost planes must be current"""
        ul = comm.halo_nodes(torch.from_numpy(u), 3).numpy()
        # elements of the local layers only; rows of owned planes are complete
        y = orc.matfree_apply(nx, ny, part.nz_local, 3, KE, El, Nl, ul)
        out = np.zeros_like(u)
        out[own] = y[own]
        return out

    def dot(a, c):
        return comm.dot_owned(torch.from_numpy(a), torch.from_numpy(c), 3)

    # ownership: every node owned exactly once
    cnt = torch.zeros(nx * ny * nz)
    cnt[part.plane * (part.node_z0 + part.own_lo): part.plane * (part.node_z0 + part.own_hi + 1)] = 1
    dist.all_reduce(cnt)
    assert bool((cnt == 1).all())
    # operator apply vs the serial oracle
    u_glob = np.random.default_rng(0).standard_normal(3 * nx * ny * nz)
    y_glob = orc.matfree_apply(nx, ny, nz, 3, KE, E, N, u_glob)
    ul = u_glob[gs].copy()
    if part.has_lo:
        ul[: 3 * part.plane] = 777.0  # stale ghosts must be overwritten by the halo exchange
    if part.has_hi:
        ul[-3 * part.plane:] = -777.0
    yl = apply_local(ul)
    assert np.abs(yl[own] - y_glob[gs][own]).max() <= 1e-13 * np.abs(y_glob).max()
    # Jacobi-preconditioned CG on slabs == serial (same arithmetic up to reduction order)
    mg = orc.MG(nx, ny, nz, 3, 1)
    mg.assemble(KE, E, N)
    dinv = 1.0 / mg.diag(0)[gs]
    xk = np.zeros_like(bl)
    r = bl.copy()
    r[own] -= apply_local(xk)[own]
    z = dinv * r
    p = z.copy()
    rz = dot(r, z)
    hist = [np.sqrt(dot(r, r))]
    for it in range(400):
        w = apply_local(p)
        alpha = rz / dot(p, w)
        xk[own] += alpha * p[own]
        r[own] -= alpha * w[own]
        hist.append(np.sqrt(dot(r, r)))
        if hist[-1] <= 1e-10 * hist[0]:
            break
        z = dinv * r
        rzn = dot(r, z)
        p[own] = z[own] + (rzn / rz) * p[own]
        rz = rzn
    import scipy.sparse.linalg as spla
    Uref = spla.spsolve(mg.csr(0).tocsc(), b)
    assert np.abs(xk[own] - Uref[gs][own]).max() <= 1e-7 * np.abs(Uref).max(), np.abs(xk[own] - Uref[gs][own]).max()
    assert comm.n_exchanges > 0 and comm.n_allreduces > 0
    # all-gather hook (replicated coarsest level): every rank sees every rank's block, in rank order
    comm.send_lo[:5] = torch.arange(5, dtype=torch.float64) + 10 * rank
    comm.allgather(5)
    want = torch.cat([torch.arange(5, dtype=torch.float64) + 10 * r for r in range(world)])
    assert torch.equal(comm.gather[: 5 * world], want)
    # multigrid level partitions are consistent
    for l in range(2):
        pl = part.level(l)
        assert pl.ez_own * world == pl.ez and pl.node_z0 == rank * pl.ez_own
    print("rank %d cpu OK its=%d" % (rank, len(hist) - 1), flush=True)


def gpu_mode(rank, world):
    import topopt_in_petsc_amd as tp
    torch.cuda.set_device(0)
    ex, ey, ez, nlv = [int(v) for v in sys.argv[2:6]] if len(sys.argv) >= 6 else (16, 8, 16, 3)
    nsm, nco = [int(v) for v in sys.argv[6:8]] if len(sys.argv) >= 8 else (4, 30)   # Chebyshev steps: smoothing, coarse solve
    cyc = [int(v) for v in sys.argv[8].split(",")] if len(sys.argv) >= 9 else None    # cycles per level (W-cycles)
    nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
    grid = tp.Grid(nx, ny, nz, h, rank=rank, nranks=world)
    part = grid.part
    # the collective self-check that guards the in-library RCCL path, here through the host hooks
    import ctypes
    ok = ctypes.c_int(0)
    assert grid.L.tp_grid_comm_selfcheck(grid.handle, ctypes.byref(ok)) == 0 and ok.value == 1
    le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=nlv, rtol=1e-9, max_it=300, nsmooth=nsm, ncoarse=nco))
    if cyc:
        le.set_cycles(cyc)
    le.SetUpLoadAndBC()
    x = grid.synth_density()
    flt = tp.Filter(grid, 1, 2.56 * h)
    xt, xp, df, dg = grid.elem_vec(), grid.elem_vec(), grid.elem_vec(), grid.elem_vec()
    flt.FilterProject(x, xt, xp)
    fx, gx = le.ComputeObjectiveConstraintsSensitivities(df, dg, xp, 1e-9, 1.0, 3.0, 0.12, hist_cap=400)
    flt.Gradients(x, xt, df, [dg])
    mnd = flt.GetMND(xp)
    xp_e = xp.clone()
    own_t = torch.arange(le.U.numel(), device="cuda")[grid.part.owned_slice(3)]
    # ---- serial oracle on the global mesh
    xo = orc.synth_density(ex, ey, ez, h)
    of = orc.Filter(nx, ny, nz, h, 2.56 * h)
    xto, xpo = of.project(1, xo)
    KE = orc.hex8_ke_box(h, h, h, 0.3)
    N, R = orc.cantilever_bc(nx, ny, nz, h)
    mg = orc.MG(nx, ny, nz, 3, nlv, nsm, nco)
    if cyc:
        mg.set_cycles(cyc)
    mg.assemble(KE, orc.simp(xpo), N)
    U, its, hist = mg.solve(R * N, rtol=1e-9, maxit=300)
    fo, go, dfo, dgo = orc.compliance_sens(nx, ny, nz, KE, U, xpo)
    dfo_f = of.gradient(1, xo, xto, dfo)
    es, own, gs = part.global_elem_slice(), part.owned_slice(3), part.global_slice(3)
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    assert rel(xp.cpu().numpy(), xpo[es]) <= 1e-13
    assert le.last_its == its, (le.last_its, its)
    hh = le.last_hist
    assert np.abs(hh[:10] / hist[:10] - 1).max() <= 1e-9
    assert np.abs(hh / hist - 1).max() <= 1e-5
    assert rel(le.U.cpu().numpy()[own], U[gs][own]) <= 1e-8
    assert abs(fx / fo - 1) <= 1e-9 and abs(gx - go) <= 1e-13
    assert rel(df.cpu().numpy(), dfo_f[es]) <= 1e-8
    assert abs(mnd - orc.mnd(xpo)) <= 1e-13
    # level operators on slabs vs the oracle's global level matrices
    rng = np.random.default_rng(7)
    for l in range(nlv):
        pl = part.level(l)
        ug = rng.standard_normal(mg.size(l))
        yg = mg.apply(l, ug)
        ul = torch.from_numpy(ug[pl.global_slice(3)].copy()).cuda()
        yl = le.level_apply(l, ul).cpu().numpy()
        assert rel(yl[pl.owned_slice(3)], yg[pl.global_slice(3)][pl.owned_slice(3)]) <= 1e-12, l
        assert abs(le.level_lambda(l) / mg.lam(l) - 1) <= 1e-9
    # PDE filter on slabs
    pf = tp.Filter(grid, 2, 2.56 * h, tp.SolverOptions(nlvls=3, rtol=1e-8, dtol=1e3, max_it=60, nsmooth=2, ncoarse=10))
    pf.FilterProject(x, xt, xp)
    opf = orc.PDEFilter(nx, ny, nz, h, 2.56 * h, nlv=3, nsmooth=2, ncoarse=10)
    xpf, its_p, _ = opf.apply(xo)
    assert pf.last_pde_solve()[0] == its_p
    assert rel(xt.cpu().numpy(), np.clip(xpf, 0, 1)[es]) <= 1e-9
    # ---- halo overlap (second stream, boundary planes first) is ON by default and changes NOTHING bitwise: the same
    # solve with TP_OVERLAP=0 semantics (a grid created with the switch off) gives the identical U and history
    assert grid.halo_overlap > 0, "no halo travelled on the second stream"
    os.environ["TP_OVERLAP"] = "0"
    grid0 = tp.Grid(nx, ny, nz, h, rank=rank, nranks=world)
    os.environ.pop("TP_OVERLAP")
    le0 = tp.LinearElasticity(grid0, tp.SolverOptions(nlvls=nlv, rtol=1e-9, max_it=300, nsmooth=nsm, ncoarse=nco))
    if cyc:
        le0.set_cycles(cyc)
    le0.SetUpLoadAndBC()
    le0.ComputeObjectiveConstraintsSensitivities(grid0.elem_vec(), grid0.elem_vec(), xp_e, 1e-9, 1.0, 3.0, 0.12, hist_cap=400)
    assert grid0.halo_overlap == 0
    assert np.array_equal(le0.last_hist, hh), "overlapped and blocking halos differ"
    assert torch.equal(le0.U[own_t], le.U[own_t])
    print("rank %d gpu OK its=%d exchanges=%d allreduces=%d overlapped=%d direct=%d allgathers=%d" %
          (rank, its, grid.comm.n_exchanges, grid.comm.n_allreduces, grid.halo_overlap, grid.comm.n_direct, grid.comm.n_allgathers), flush=True)


def gpu_randbc_mode(rank, world):
    """randomly scattered Dirichlet dofs on slabs: flagged level-1 elements on both sides of every slab boundary (compact
    ghost rows of the correction, masks on ghost planes) -- level operators and the solve against the serial oracle"""
    import topopt_in_petsc_amd as tp
    torch.cuda.set_device(0)
    ex, ey, ez, nlv = [int(v) for v in sys.argv[2:6]] if len(sys.argv) >= 6 else (16, 8, 16, 3)
    nsm, nco = [int(v) for v in sys.argv[6:8]] if len(sys.argv) >= 8 else (4, 30)   # Chebyshev steps: smoothing, coarse solve
    cyc = [int(v) for v in sys.argv[8].split(",")] if len(sys.argv) >= 9 else None    # cycles per level (W-cycles)
    nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
    rng = np.random.default_rng(11)
    N = np.ones(3 * nx * ny * nz)
    N[rng.random(N.size) < 0.03] = 0.0
    for n in rng.integers(0, nx * ny * nz, size=40):
        N[3 * n: 3 * n + 3] = 0.0
    N[: 3 * nx] = 0.0
    R = rng.standard_normal(N.size) * 1e-3
    grid = tp.Grid(nx, ny, nz, h, rank=rank, nranks=world)
    part = grid.part
    gs, own = part.global_slice(3), part.owned_slice(3)
    le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=nlv, rtol=1e-9, max_it=400))
    le.SetBC(torch.from_numpy(N[gs].copy()).cuda(), torch.from_numpy(R[gs].copy()).cuda())
    xo = orc.synth_density(ex, ey, ez, h)
    le.AssembleStiffnessMatrix(torch.from_numpy(xo[part.global_elem_slice()].copy()).cuda(), 1e-9, 1.0, 3.0)
    KE = orc.hex8_ke_box(h, h, h, 0.3)
    mg = orc.MG(nx, ny, nz, 3, nlv)
    mg.assemble(KE, orc.simp(xo), N)
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    for l in range(nlv):
        pl = part.level(l)
        ug = rng.standard_normal(mg.size(l))
        yg = mg.apply(l, ug)
        yl = le.level_apply(l, torch.from_numpy(ug[pl.global_slice(3)].copy()).cuda()).cpu().numpy()
        assert rel(yl[pl.owned_slice(3)], yg[pl.global_slice(3)][pl.owned_slice(3)]) <= 1e-12, l
        assert abs(le.level_lambda(l) / mg.lam(l) - 1) <= 1e-9
    its = le.KSPSolve(hist_cap=400)
    U, its_o, hist = mg.solve(R * N, rtol=1e-9, maxit=400)
    assert its == its_o, (its, its_o)
    assert np.abs(le.last_hist[:10] / hist[:10] - 1).max() <= 1e-8
    assert rel(le.U.cpu().numpy()[own], U[gs][own]) <= 1e-7
    print("rank %d gpu_randbc OK its=%d" % (rank, its), flush=True)


def gpu_giveup_mode(rank, world):
    """a one-XCD kernel that gives up on ONE rank (TP_TEST_FORCE_GIVEUP="<1 solve | 2 set-up>:<rank>", set by the
    test) -- every rank must take the recovery branch (hierarchy rebuilt without the one-XCD forms, solve repeated): the run
    ends on all ranks, the replicated coarse solve is the same kind everywhere afterwards, the result is the oracle's."""
    import topopt_in_petsc_amd as tp
    torch.cuda.set_device(0)
    ex, ey, ez, nlv, nsm, nco = 32, 16, 64, 4, 2, 20   # coarsest level 5 x 3 x 9 nodes = 405 rows: replicated, solved exactly (coarse_direct = 2)
    nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
    grid = tp.Grid(nx, ny, nz, h, rank=rank, nranks=world)
    le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=nlv, rtol=1e-9, max_it=300, nsmooth=nsm, ncoarse=nco, coarse_direct=2))
    le.SetUpLoadAndBC()
    x = grid.synth_density()
    df, dg = grid.elem_vec(), grid.elem_vec()
    fx, gx = le.ComputeObjectiveConstraintsSensitivities(df, dg, x, 1e-9, 1.0, 3.0, 0.12, hist_cap=400)
    direct_after = int(le.coarse_direct_active())
    t = torch.tensor([direct_after], dtype=torch.int64)
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    assert all(int(v) == 0 for v in gathered), "after the recovery the exact coarse solve must be off on EVERY rank: %s" % gathered
    xo = orc.synth_density(ex, ey, ez, h)
    KE = orc.hex8_ke_box(h, h, h, 0.3)
    N, R = orc.cantilever_bc(nx, ny, nz, h)
    mg = orc.MG(nx, ny, nz, 3, nlv, nsm, nco)
    mg.assemble(KE, orc.simp(xo), N)
    U, its, hist = mg.solve(R * N, rtol=1e-9, maxit=300)
    fo = orc.compliance_sens(nx, ny, nz, KE, U, xo)[0]
    assert le.last_its == its, (le.last_its, its)
    assert abs(fx / fo - 1) <= 1e-8
    print("rank %d gpu_giveup OK its=%d" % (rank, its), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    try:
        {"cpu": cpu_mode, "gpu": gpu_mode, "gpu_randbc": gpu_randbc_mode, "gpu_giveup": gpu_giveup_mode}[mode](rank, world)
    finally:
        dist.destroy_process_group()

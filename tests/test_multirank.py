"""N > 1 path: world_size-2 runs.  CPU (gloo): partition + halo + reductions with an
oracle-driven slab CG.  GPU: the real HIP solver on two z-slabs sharing one GPU
(gloo with host staging) against the serial oracle."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(mode, nproc=2, timeout=600, extra=(), env_extra=None):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "mp_gloo_worker.py"), mode] + [str(v) for v in extra]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    for k in range(nproc):
        assert "rank %d %s OK" % (k, mode) in r.stdout, r.stdout[-2000:]
    return r.stdout


def test_partition_logic():
    from topopt_in_petsc_amd.partition import SlabPartition
    nx, ny, nz = 9, 5, 17
    owned = np.zeros(nz, dtype=int)
    for r in range(4):
        p = SlabPartition(nx, ny, nz, r, 4)
        assert p.ez_own == 4 and p.elem_z0 == 4 * r and p.node_z0 == 4 * r
        assert p.nz_local == (6 if r < 3 else 5)
        assert (p.own_lo, p.own_hi) == ((0, 4) if r == 0 else (1, 4))
        owned[p.node_z0 + p.own_lo: p.node_z0 + p.own_hi + 1] += 1
        assert p.n_owned_nodes == nx * ny * (5 if r == 0 else 4)
        assert p.coarsenable(3) and not p.coarsenable(4)
        assert p.level(2).ez_own == 1
    assert (owned == 1).all()
    with pytest.raises(ValueError):
        SlabPartition(9, 5, 16, 0, 4)


def test_two_ranks_cpu_gloo():
    _launch("cpu")


@pytest.mark.gpu
def test_two_ranks_one_gpu():
    _launch("gpu")


@pytest.mark.gpu
def test_four_ranks_one_gpu_interior_slabs():
    """4 slabs on one GPU: the two interior ranks have a neighbour on both sides (ghost plane below AND above),
    on a mesh that does not line up with the tile sizes"""
    _launch("gpu", nproc=4, extra=(20, 12, 32, 3))



@pytest.mark.gpu
def test_bench_cycle_on_slabs():
    """the V-cycle bench.py runs on the metric mesh (5 levels, Chebyshev(2) smoothing, Chebyshev(45) coarse solve) on two
    slabs: four levels stay distributed, the fifth is the replicated copy"""
    _launch("gpu", nproc=2, extra=(32, 16, 64, 5, 2, 45))


@pytest.mark.gpu
def test_bench_w_cycle_on_slabs():
    """the cycle of the bench's metric mesh since the W-cycle scan (5 levels, Chebyshev(2), coarse Chebyshev(20), levels 2 and 3
    cycled twice) on two slabs, against the serial oracle; overlapped and blocking halos bit-equal"""
    _launch("gpu", nproc=2, extra=(32, 16, 64, 5, 2, 20, "1,2,2,1"))


@pytest.mark.gpu
def test_eight_ranks_one_gpu_c3_c5_slab_geometry():
    """The 8-GPU slab geometry of BASELINE configs C3 (256x128x128) and C5 (512x256x256) before hardware sees it:
    8 ranks, 4 multigrid levels, 16 fine = 2 coarsest-level element layers per rank, replicated coarsest level,
    interior ranks with neighbours on both sides -- on a mesh reduced in x-y (16x8x128), every rank on cuda:0,
    against the serial oracle (residual history, U, objective, sensitivities, filters, every level operator)."""
    _launch("gpu", nproc=8, timeout=900, extra=(16, 8, 128, 4))


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,mesh", [(2, (16, 8, 16, 3)), (4, (24, 12, 32, 3))])
def test_random_dirichlet_on_slabs(nproc, mesh):
    """Dirichlet dofs scattered at random across the slab boundaries (flagged level-1 elements in the ghost layers,
    masks on ghost planes): slabs on one GPU against the serial oracle"""
    _launch("gpu_randbc", nproc=nproc, extra=mesh)


@pytest.mark.gpu
def test_two_levels_on_slabs_coarsest_level_stays_distributed():
    """two multigrid levels on 2 slabs: the coarsest level is level 1 (applied from the fine moduli, nothing to
    replicate) and keeps its halo exchanges -- against the serial oracle"""
    _launch("gpu", nproc=2, extra=(16, 8, 8, 2))


@pytest.mark.gpu
def test_too_thin_slabs_are_refused_on_every_rank():
    """one element layer per rank on a distributed level: refused at creation with TP_ERR_ARG on all ranks alike (a
    rank-dependent failure later would leave the others waiting in a collective)"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "mp_gloo_worker.py"), "gpu", "16", "8", "8", "2"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=180)
    assert r.returncode != 0
    assert (r.stdout + r.stderr).count("too few for 2 multigrid levels on slabs") == 4
    assert "tp_elasticity_create failed: TP_ERR_ARG" in r.stdout + r.stderr


@pytest.mark.gpu
def test_three_ranks_one_gpu_odd_rank_count():
    """an odd number of slabs (the middle rank has neighbours on both sides, the replicated coarsest level is gathered
    from three unequal parts: rank 0 owns one node plane more)"""
    _launch("gpu", nproc=3, extra=(16, 8, 24, 3))


@pytest.mark.gpu
@pytest.mark.parametrize("force", ["1:1", "2:0"])
def test_give_up_on_one_rank_is_handled_by_all(force):
    """the one-XCD give-up is detected per rank, the recovery is collective.  One rank of two takes the
    forced give-up branch (1: found by the solve, on rank 1; 2: found by the set-up, on rank 0): both ranks rebuild and
    finish, with the launch-per-step coarse solve on both, and the oracle's iteration count and compliance."""
    _launch("gpu_giveup", nproc=2, timeout=300, env_extra={"TP_TEST_FORCE_GIVEUP": force})


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,mesh,env", [
    (2, (32, 16, 64, 5, 2, 20, "1,3,1,1"), {"TP_REPLICATE_FROM": "2"}),     # levels 2, 3, 4 on every rank (forced)
    (4, (32, 16, 64, 5, 2, 20, "1,3,1,1"), {}),                            # 16 layers per rank: level 3 holds 2 -> levels 3, 4 (automatic)
    (2, (32, 16, 64, 5, 2, 20, "1,3,1,1"), {"TP_REPLICATE_FROM": "0"}),     # the coarsest level only (round 2's form)
])
def test_coarse_levels_replicated_on_every_rank(nproc, mesh, env):
    """Round 5 (VERDICT r4 'next' 7): the coarse levels from rep0 on as replicated global copies -- one all-gather of the
    right-hand side per visit, no halo exchange on those levels -- against the serial oracle (iteration count, history, U,
    objective, sensitivities, every level operator and eigenvalue estimate; overlapped = blocking halos bitwise), with the
    bench's cycle pattern; the three ways rep0 is chosen."""
    _launch("gpu", nproc=nproc, timeout=600, extra=mesh, env_extra=env)


@pytest.mark.gpu
def test_replicated_levels_take_their_halo_exchanges_out():
    """the point of the replicated sub-hierarchy: with levels 2-4 on every rank the solve issues far fewer halo exchanges than
    with the coarsest level only (same mesh, same cycle, same iteration count -- both runs are checked against the oracle)"""
    import re
    mesh = (32, 16, 64, 5, 2, 20, "1,3,1,1")
    n = {}
    for rep in ("0", "2"):
        out = _launch("gpu", nproc=2, timeout=600, extra=mesh, env_extra={"TP_REPLICATE_FROM": rep})
        m = re.search(r"rank 0 gpu OK its=(\d+) exchanges=(\d+) allreduces=(\d+) overlapped=(\d+) direct=(\d+) allgathers=(\d+)", out)
        n[rep] = (int(m.group(1)), int(m.group(2)) + int(m.group(5)), int(m.group(6)))    # its, halo exchanges (staged + in place), all-gathers
    assert n["0"][0] == n["2"][0]
    assert n["2"][1] < 0.7 * n["0"][1], n     # measured: 979 -> 419 halo exchanges (and 48 -> 22 all-gathers: one per visit of level 2
                                              # instead of one per visit of the coarsest level)

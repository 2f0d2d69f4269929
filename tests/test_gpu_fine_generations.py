"""The three generations of the fine-level operator kernel (csrc/matfree_tile.h, fine_tile.h, fine_u4.h) must give the
same BITS: product, fused Chebyshev steps (zero and non-zero guess), V-cycle and a whole solve, on a cantilever and on
scattered Dirichlet sets, for both tile shapes of the third generation.  The generation is read from the environment
once per process, hence one subprocess per setting."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
import topopt_in_petsc_amd as tp
tp.load_library()
ex, ey, ez, scattered, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
grid = tp.Grid(nx, ny, nz, h)
le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=3, rtol=1e-8))
rng = np.random.default_rng(7)
if scattered:
    N = np.ones(3 * nx * ny * nz)
    N[rng.random(N.size) < 2e-3] = 0.0
    N[: 3 * nx].reshape(-1, 3)[:, :] = 0.0
    R = rng.standard_normal(N.size) * 1e-3
    le.SetBC(torch.from_numpy(N).cuda(), torch.from_numpy(R).cuda())
else:
    le.SetUpLoadAndBC()
le.AssembleStiffnessMatrix(grid.synth_density(12345), 1e-9, 1.0, 3.0)
u = torch.from_numpy(rng.standard_normal(3 * nx * ny * nz)).cuda()
b = torch.from_numpy(rng.standard_normal(3 * nx * ny * nz)).cuda()
res = {}
res["apply"] = le.MatMult(u).cpu().numpy()
x = torch.zeros_like(u)
le.smooth(0, b, x, 3, True)
res["cheb0"] = x.cpu().numpy()
x = u.clone()
le.smooth(0, b, x, 4, False)
res["cheb1"] = x.cpu().numpy()
res["pc"] = le.precond(b).cpu().numpy()
its = le.KSPSolve(hist_cap=300)
res["U"] = le.U.cpu().numpy()
res["hist"] = np.asarray(le.last_hist)
res["its"] = np.asarray([its])
np.savez(out, **res)
"""


def run(tmp_path, tag, env, mesh, scattered):
    out = str(tmp_path / ("%s.npz" % tag))
    e = dict(os.environ)
    for k in ("TP_FINE_V", "TP_FINE_SHAPE", "TP_TILE_KZ"):
        e.pop(k, None)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}] + [str(v) for v in mesh] + [str(int(scattered)), out], env=e,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


@pytest.mark.parametrize("mesh,scattered", [((40, 24, 20), False), ((36, 28, 20), True), ((64, 32, 32), False)])
def test_generations_bitwise(tmp_path, mesh, scattered):
    ref = run(tmp_path, "g2", {"TP_FINE_V": "2"}, mesh, scattered)
    settings = {"g1": {"TP_FINE_V": "1"}, "g3_16x16": {"TP_FINE_V": "3", "TP_FINE_SHAPE": "1"},
                "g3_32x8": {"TP_FINE_V": "3", "TP_FINE_SHAPE": "2"}, "g3_32x8_kz3": {"TP_FINE_V": "3", "TP_FINE_SHAPE": "2", "TP_TILE_KZ": "3"},
                "auto": {}}
    for tag, env in settings.items():
        got = run(tmp_path, tag, env, mesh, scattered)
        assert int(got["its"][0]) == int(ref["its"][0]), tag
        for k in ("apply", "cheb0", "cheb1"):
            # generation 1 is compiled with implicit contraction and has another Chebyshev epilogue: bits only from 2 on
            if tag == "g1":
                assert np.abs(got[k] - ref[k]).max() <= 1e-12 * np.abs(ref[k]).max(), (tag, k)
            else:
                assert np.array_equal(got[k].view(np.int64), ref[k].view(np.int64)), (tag, k)
        if tag != "g1":
            assert np.array_equal(got["pc"].view(np.int64), ref["pc"].view(np.int64)), (tag, "pc")
            # the solve also depends on the dot products fused into the kernels: one partial sum per workgroup, i.e. a
            # summation order that follows the tile shape and the z-chunks -> last-bit differences in alpha, beta
            for k in ("U", "hist"):
                assert np.abs(got[k] - ref[k]).max() <= 1e-9 * np.abs(ref[k]).max(), (tag, k)


STENCIL_WORKER = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
import topopt_in_petsc_amd as tp
tp.load_library()
ex, ey, ez, nlv, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
grid = tp.Grid(ex + 1, ey + 1, ez + 1, 1.0 / ey)
le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=nlv, nsmooth=2, ncoarse=20, rtol=1e-8, coarse_direct=1))
le.set_cycles([1, 3, 1, 1][: nlv - 1])
le.SetUpLoadAndBC()
le.AssembleStiffnessMatrix(grid.synth_density(12345), 1e-9, 1.0, 3.0)
rng = np.random.default_rng(11)
res = {}
for l in range(1 if os.environ.get("TP_NO_MACRO") else 2, nlv):
    n = 3 * le.level_nodes(l)
    u, b = torch.from_numpy(rng.standard_normal(n)).cuda(), torch.from_numpy(rng.standard_normal(n)).cuda()
    res["apply%%d" %% l] = le.level_apply(l, u).cpu().numpy()
    x = torch.zeros_like(u)
    le.smooth(l, b, x, 3, True)
    res["cheb0_%%d" %% l] = x.cpu().numpy()
    x = u.clone()
    le.smooth(l, b, x, 4, False)
    res["cheb1_%%d" %% l] = x.cpu().numpy()
    res["lam%%d" %% l] = np.asarray([le.level_lambda(l)])
res["lam_level1"] = np.asarray([le.level_lambda(1)])
r = torch.from_numpy(rng.standard_normal(3 * (ex + 1) * (ey + 1) * (ez + 1))).cuda()
res["pc"] = le.precond(r).cpu().numpy()
its = le.KSPSolve(hist_cap=300)
res["U"], res["hist"], res["its"] = le.U.cpu().numpy(), np.asarray(le.last_hist), np.asarray([its])
np.savez(out, **res)
"""


# level 2 of the first three: 18 513 / 15 625 / 14 025 nodes; the fourth stores level 1 as a stencil (TP_NO_MACRO: 65^3 = 274 625
# nodes, the size of the 256^3 class's level 2, where the row form used to run unsplit: both forms three-way split here)
@pytest.mark.parametrize("mesh,nlv,extra", [((128, 128, 64), 4, {}), ((96, 96, 96), 5, {}), ((128, 96, 64), 4, {}),
                                            ((128, 128, 128), 3, {"TP_NO_MACRO": "1", "TP_DIA_SPLIT": "3"})])
def test_stencil_kernel_per_node_equals_per_row_bitwise(tmp_path, mesh, nlv, extra):
    """Round 6: on the stored-stencil levels whose rows are split three ways, a thread per NODE and z-offset (k_dia_node3: a third
    of the waves, the index arithmetic of a node shared by its three rows) against a thread per ROW and z-offset
    (k_dia_row_split<3, EPI, 3>; TP_DIA_NODE=0): operator, Chebyshev steps from a zero and a non-zero guess, the eigenvalue
    estimates, one V-cycle and a whole solve -- the same bits (every row is summed in the same order)."""
    res = {}
    for tag, env in (("node", {}), ("row", {"TP_DIA_NODE": "0"})):
        out = str(tmp_path / (tag + ".npz"))
        e = dict(os.environ)
        for k in ("TP_DIA_NODE", "TP_NO_MACRO", "TP_NO_DIA_SYM", "TP_DIA_SPLIT"):
            e.pop(k, None)
        e.update(env)
        e.update(extra)
        r = subprocess.run([sys.executable, "-c", STENCIL_WORKER % {"root": ROOT}] + [str(v) for v in mesh] + [str(nlv), out],
                           env=e, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = np.load(out)
    a, b = res["node"], res["row"]
    assert sorted(a.files) == sorted(b.files) and any(k.startswith("apply") for k in a.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("mesh,nlv", [((96, 96, 96), 5), ((128, 96, 64), 4)])
def test_spectra_chains_with_in_kernel_reductions_against_the_plain_chains(tmp_path, mesh, nlv):
    """Round 6: the Lanczos chains of the set-up end their dot products inside k_multi_dot (same order as k_reduce_multi: the
    same bits), take beta from the second Gram-Schmidt subtraction (another summation order than a dot product of its own)
    and scale inside the level-1 operator ((y + c) d became y d + c d at the Dirichlet corrections); TP_LANCZOS_TAILS=0 keeps
    the chain of round 5, TP_LANCZOS_ON_MAIN=0 its streams.  The estimates agree to rounding -- far inside what the
    residual-history parity needs -- and the solve takes the same iterations."""
    res = {}
    for tag, env in (("tails", {}), ("plain", {"TP_LANCZOS_TAILS": "0", "TP_LANCZOS_ON_MAIN": "0"})):
        out = str(tmp_path / (tag + ".npz"))
        e = dict(os.environ)
        for k in ("TP_LANCZOS_TAILS", "TP_LANCZOS_ON_MAIN", "TP_NO_REDUCE_TAIL"):
            e.pop(k, None)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", STENCIL_WORKER % {"root": ROOT}] + [str(v) for v in mesh] + [str(nlv), out],
                           env=e, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = np.load(out)
    a, b = res["tails"], res["plain"]
    lams = [k for k in a.files if k.startswith("lam")]
    assert lams
    for k in lams:
        assert abs(a[k][0] - b[k][0]) <= 1e-12 * abs(b[k][0]), (k, a[k][0], b[k][0])
    assert int(a["its"][0]) == int(b["its"][0])
    n = int(a["its"][0])
    assert np.abs(a["hist"][:n + 1] - b["hist"][:n + 1]).max() <= 1e-10 * np.abs(b["hist"][:n + 1]).max()
    assert np.abs(a["U"] - b["U"]).max() <= 1e-9 * np.abs(b["U"]).max()
    for k in a.files:   # the operators themselves do not depend on the chains
        if k.startswith("apply"):
            assert np.array_equal(a[k], b[k]), k


def test_large_level_node_stencil_with_mirrored_reads_against_the_plain_form(tmp_path):
    """on a large stencil level (65^3 nodes: the 256^3 class's level 2, stored here as level 1 of 128^3) the node form reads the
    upper half of the stencil from the neighbours' rows (mirrored, transposed blocks: half the coefficient stream -- what took
    that level from 121 to 73 us per Chebyshev step); against the same kernel reading every coefficient from its own row
    (TP_NO_DIA_SYM): operator, smoothing steps, V-cycle and solve agree to rounding (the stored stencil is symmetric to the
    rounding of its Galerkin sums)"""
    res = {}
    mesh, nlv = (128, 128, 128), 3
    for tag, env in (("sym", {}), ("plain", {"TP_NO_DIA_SYM": "1"})):
        out = str(tmp_path / (tag + ".npz"))
        e = dict(os.environ)
        for k in ("TP_DIA_NODE", "TP_NO_DIA_SYM"):
            e.pop(k, None)
        e.update(env)
        e["TP_NO_MACRO"] = "1"
        r = subprocess.run([sys.executable, "-c", STENCIL_WORKER % {"root": ROOT}] + [str(v) for v in mesh] + [str(nlv), out],
                           env=e, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = np.load(out)
    a, b = res["sym"], res["plain"]
    assert int(a["its"][0]) == int(b["its"][0])
    for k in a.files:
        if k.startswith(("apply", "cheb", "pc")):
            assert np.abs(a[k] - b[k]).max() <= 1e-12 * np.abs(b[k]).max(), k
    assert np.abs(a["U"] - b["U"]).max() <= 1e-9 * np.abs(b["U"]).max()      # (a solve to rtol 1e-8 under a preconditioner that differs in its last bits)

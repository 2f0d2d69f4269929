"""Acceptance of the PETSc-named boundary (include/petsc_compat/petsc.h).

CPU part (build container, where /root/reference exists): the reference's LinearElasticity.cc, Filter.cc and
PDEFilter.cc compile UNCHANGED against the compat headers and link against libtopopt_petsc_shim.so
(host/build_ref_on_shim.sh; nothing of the reference is stored in the repository, the binaries are git-ignored
build artefacts) -- and so do main.cc, TopOpt.cc, MMA.cc and MPIIO.cc: the reference's WHOLE program.  GPU part: that binary -- the reference's own classes on the MI355X path -- against the product's
Python API on the same mesh; with no solver options at all the reference's hard-coded FGMRES/GMRES/SOR configuration
runs as written (a one-device correctness mode), anything else is refused."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "host", "_refbuild", "ref_on_shim")
OPTS = ("-ksp_type cg -mg_levels_ksp_type chebyshev -mg_levels_pc_type jacobi "
        "-mg_coarse_ksp_type chebyshev -mg_coarse_pc_type jacobi").split()


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference sources only exist in the build container")
def test_reference_sources_compile_and_link_against_the_shim():
    r = subprocess.run(["bash", os.path.join(ROOT, "host", "build_ref_on_shim.sh")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert os.path.exists(BIN) and os.path.exists(os.path.join(ROOT, "host", "_refbuild", "topopt_ref"))
    # every PETSc symbol the three objects need is exported by the shim (the link above would have failed otherwise);
    # and the shim's header declares nothing it does not define
    src = open(os.path.join(ROOT, "include", "petsc_compat", "petsc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b((?:Petsc|Vec|Mat|KSP|PC|DM|MPI_)[A-Za-z0-9_]*)\s*\(", src))
    names -= {"PetscMalloc", "PetscFree", "PetscMin", "PetscMax", "PetscAbsScalar", "PetscAbsReal", "PetscSqrtScalar",
              "PetscSqrtReal", "PetscPowScalar", "PetscPowReal", "PetscRealPart"}
    so = os.path.join(ROOT, "topopt_in_petsc_amd", "libtopopt_petsc_shim.so")
    exported = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    missing = [n for n in sorted(names) if not re.search(r"\b%s\b" % n, exported)]
    assert not missing, missing


def _run(args, env=None):
    return subprocess.run([BIN] + [str(a) for a in args], capture_output=True, text=True, timeout=300,
                          env=dict(os.environ, **(env or {})))


def _numbers(out):
    m = re.search(r"REF_ON_SHIM fx (\S+) gx (\S+) sum_dfdx (\S+) sum_dgdx (\S+) sum_xphys (\S+) normU (\S+)", out)
    assert m, out[-2000:]
    return [float(v) for v in m.groups()]


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(BIN), reason="host/_refbuild/ref_on_shim not built (build container only)")
@pytest.mark.parametrize("ftype", [1, 2])
def test_reference_classes_on_the_mi355x_path(ftype):
    import torch
    import topopt_in_petsc_amd as tp
    ex, ey, ez = 32, 16, 16
    r = _run([ex, ey, ez, ftype] + OPTS)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    fx, gx, sdf, sdg, sxp, un = _numbers(r.stdout)
    its = int(re.search(r"State solver:\s+iter: (\d+)", r.stdout).group(1))
    # ---- the product's own API, same mesh / defaults (LinearElasticity.cc:22-23, :621-635; PDEFilter.cc:32, :280-283)
    h = 1.0 / ey
    grid = tp.Grid(ex + 1, ey + 1, ez + 1, h)
    le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=4))
    le.SetUpLoadAndBC()
    popt = tp.SolverOptions(nlvls=3, rtol=1e-8, dtol=1e3, max_it=60, nsmooth=1, ncoarse=10) if ftype == 2 else None   # :371-378 one smoothing step
    flt = tp.Filter(grid, ftype, 2.56 * h, popt)
    x = grid.synth_density(12345)
    xt, xp, df, dg = grid.elem_vec(), grid.elem_vec(), grid.elem_vec(), grid.elem_vec()
    flt.FilterProject(x, xt, xp)
    fx2, gx2 = le.ComputeObjectiveConstraintsSensitivities(df, dg, xp, 1e-9, 1.0, 3.0, 0.12)
    flt.Gradients(x, xt, df, [dg])
    tol = 1e-8 if ftype == 1 else 1e-6   # type 2: two iterative Helmholtz solves sit in between
    assert its == le.last_its
    assert sxp == pytest.approx(float(xp.sum()), rel=tol)
    assert fx == pytest.approx(fx2, rel=tol) and gx == pytest.approx(gx2, abs=tol)
    assert sdf == pytest.approx(float(df.sum()), rel=tol) and sdg == pytest.approx(float(dg.sum()), rel=tol)
    assert un == pytest.approx(float(le.U.norm()), rel=tol)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(BIN), reason="host/_refbuild/ref_on_shim not built (build container only)")
@pytest.mark.parametrize("ftype", [1, 2])
def test_reference_hard_coded_solver_runs_as_written(ftype):
    """No solver options at all: the reference's SetUpSolver bodies configure FGMRES(100) + PCMG with GMRES(4)/SOR
    smoothers and a GMRES(30)/SOR coarse solve (LinearElasticity.cc:638, :720-746), PDEFilt's FGMRES(20) with
    GMRES(1)/Jacobi (PDEFilter.cc:276-378).  The shim runs exactly that (csrc/refksp.h, a one-device correctness mode);
    the numbers agree with the option-selected fast configuration to the solver tolerances, and the iteration count of the
    state solve equals the CPU restatement's (oracle/refksp.py)."""
    from oracle import oracle as orc, refksp
    ex, ey, ez = 16, 8, 8
    r = _run([ex, ey, ez, ftype, "-nlvls", "3"])
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "fgmres" in r.stdout                      # the reference's own settings print (:758-780)
    a = _numbers(r.stdout)
    its = int(re.search(r"State solver:\s+iter: (\d+)", r.stdout).group(1))
    r2 = _run([ex, ey, ez, ftype, "-nlvls", "3"] + OPTS)
    assert r2.returncode == 0, r2.stdout[-1500:] + r2.stderr[-1500:]
    b = _numbers(r2.stdout)
    its_cg = int(re.search(r"State solver:\s+iter: (\d+)", r2.stdout).group(1))
    assert its < its_cg
    # fx, gx, sum dfdx, sum dgdx, sum xPhys, |U|: both state solves stop at rtol 1e-5 -> agreement ~1e-4, not more
    assert a == pytest.approx(b, rel=2e-4, abs=1e-9)
    if ftype == 1:                                   # the state solve of the restatement on the same filtered density
        import torch
        import topopt_in_petsc_amd as tp
        h = 1.0 / ey
        grid = tp.Grid(ex + 1, ey + 1, ez + 1, h)
        flt = tp.Filter(grid, 1, 2.56 * h)
        x = grid.synth_density(12345)
        xt, xp = grid.elem_vec(), grid.elem_vec()
        flt.FilterProject(x, xt, xp)
        KE = orc.hex8_ke_box(h, h, h, 0.3)
        N, R = orc.cantilever_bc(ex + 1, ey + 1, ez + 1, h)
        mg = orc.MG(ex + 1, ey + 1, ez + 1, 3, 3)
        mg.assemble(KE, orc.simp(xp.cpu().numpy()), N)
        assert refksp.RefSolver(mg).solve(R * N)[1] == its


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(BIN), reason="host/_refbuild/ref_on_shim not built (build container only)")
def test_unsupported_solver_configurations_are_refused_not_substituted():
    """anything but the two implemented configurations: PETSC_ERR_SUP = 56 with a message, never a silent substitute"""
    for extra, word in (["-ksp_type", "gmres"], "gmres"), (["-mg_levels_ksp_type", "richardson"], "richardson"), \
            (["-ksp_type", "cg"], "gmres/sor"):      # CG outside, the hard-coded GMRES/SOR inside: not one of the two
        r = _run([16, 8, 8, 1, "-nlvls", "3"] + extra)
        assert r.returncode != 0 and "PETSC_ERR_SUP" in r.stderr and word in r.stderr, r.stderr[-800:]
        assert "REF_ON_SHIM failed: 56" in r.stdout


TOPOPT_REF = os.path.join(ROOT, "host", "_refbuild", "topopt_ref")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(TOPOPT_REF), reason="host/_refbuild/topopt_ref not built (build container only)")
def test_whole_reference_program_unchanged_on_the_gpu_path(tmp_path):
    """ALL eight sources of the reference (main.cc, TopOpt.cc, MMA.cc, MPIIO.cc, LinearElasticity.cc, Filter.cc,
    PDEFilter.cc), compiled unchanged against include/petsc_compat and linked against the shim, run the
    optimisation loop on the MI355X path; the product's own driver (device MMA) produces the same history.
    The reference prints 6 decimals: that is the comparison."""
    import topopt_in_petsc_amd as tp
    nit = 6
    r = subprocess.run([TOPOPT_REF, "-nx", "65", "-ny", "33", "-nz", "33", "-maxItr", str(nit)] + OPTS, capture_output=True,
                       text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    ref = [[float(v) for v in m] for m in re.findall(
        r"It\.: \d+, True fx: (\S+), Scaled fx: (\S+), gx\[0\]: (\S+), ch\.: (\S+), mnd\.: (\S+),", r.stdout)]
    its = [int(v) for v in re.findall(r"State solver:\s+iter: (\d+)", r.stdout)]
    assert len(ref) == nit and len(its) == nit
    ex, ey, ez, nlv = 64, 32, 32, 4
    h = 1.0 / ey
    opt = tp.TopOpt(nxyz=(ex + 1, ey + 1, ez + 1), xc=(0, ex * h, 0, 1, 0, ez * h), nlvls=nlv, rmin=2.56 * h, filter=1,
                    solver=tp.SolverOptions(nlvls=nlv))
    for it in range(nit):
        rec = opt.step()
        fx, sfx, gx, ch, mnd = ref[it]
        assert rec["ksp_its"] == its[it]
        assert fx == pytest.approx(rec["fx"], rel=2e-6, abs=2e-6)
        assert gx == pytest.approx(rec["gx"], abs=2e-6)
        assert ch == pytest.approx(rec["ch"], abs=2e-6)
        assert mnd == pytest.approx(rec["mnd"], abs=2e-6)
    # the design the REFERENCE's own MMA.cc produced (its restart file: x, xPhys, xo1, xo2, U, L after the last
    # iteration) against the product's device MMA after the same number of steps: the update itself, not only the
    # printed scalars
    rv = _petsc_vecs(os.path.join(str(tmp_path), "Restart00.dat"))
    assert len(rv) == 6
    x_ref, xphys_ref, xo1_ref = rv[0], rv[1], rv[2]
    x_dev = opt.x.cpu().numpy()
    assert np.abs(x_ref - x_dev).max() <= 1e-9, np.abs(x_ref - x_dev).max()   # measured: 3e-12 after 6 iterations
    assert np.abs(xphys_ref - opt.xPhys.cpu().numpy()).max() <= 1e-9
    # the result container the reference's MPIIO wrote through the compat MPI-IO has the documented layout
    out = os.path.join(str(tmp_path), "output_00000.dat")
    assert os.path.exists(out) and open(out, "rb").read(26) == b"TopOpt result version 1.1\n"


def _petsc_vecs(path):
    """all Vecs of a PETSc binary file: big-endian (classid 1211214, n, n doubles)"""
    raw = open(path, "rb").read()
    out, at = [], 0
    while at < len(raw):
        cls, n = np.frombuffer(raw, dtype=">i4", count=2, offset=at)
        assert cls == 1211214
        out.append(np.frombuffer(raw, dtype=">f8", count=int(n), offset=at + 8).astype(np.float64))
        at += 8 + 8 * int(n)
    return out


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(TOPOPT_REF), reason="host/_refbuild/topopt_ref not built (build container only)")
def test_whole_reference_program_on_slab_ranks(tmp_path):
    """The reference's whole program, unchanged, as 2 and 4 processes (host/slabrun; the compat layer's MPI is the
    shared-memory job of host/slab_comm.h, its DMDA a 1 x 1 x R process grid of z-slabs with PETSc's ownership and
    ghost ranges, its Vecs the library's slab arrays): same optimisation history as one process, the same restart
    vectors (design, MMA history, state) in natural ordering, and a result container."""
    run = os.path.join(ROOT, "host", "slabrun")
    nit, hist, vecs = 3, {}, {}
    for n in (1, 2, 4):
        wd = tmp_path / ("r%d" % n)
        wd.mkdir()
        r = subprocess.run([run, "-n", str(n), "--same-device", TOPOPT_REF, "-nx", "65", "-ny", "33", "-nz", "33", "-maxItr", str(nit)] + OPTS,
                           capture_output=True, text=True, timeout=600, cwd=str(wd))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        lines = [re.sub(r"time: .*", "", ln) for ln in r.stdout.splitlines() if ln.startswith(("It.:", "State solver"))]
        assert len(lines) == 2 * nit, r.stdout[-3000:]
        hist[n] = [[float(v) for v in re.findall(r"-?\d+\.?\d*(?:[eE][+-]?\d+)?", ln)] for ln in lines]
        vecs[n] = _petsc_vecs(str(wd / "Restart00.dat")) + _petsc_vecs(str(wd / "RestartSol00.dat"))
        assert open(str(wd / "output_00000.dat"), "rb").read(26) == b"TopOpt result version 1.1\n"
    for n in (2, 4):
        for a, b in zip(hist[n], hist[1]):
            assert a == pytest.approx(b, rel=2e-5, abs=2e-6), (n, a, b)
        assert len(vecs[n]) == len(vecs[1]) and len(vecs[1]) >= 4
        for a, b in zip(vecs[n], vecs[1]):
            assert a.shape == b.shape
            assert np.abs(a - b).max() <= 1e-7 * max(np.abs(b).max(), 1e-30), n


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(BIN), reason="host/_refbuild/ref_on_shim not built (build container only)")
@pytest.mark.parametrize("filt", [1, 2])
def test_reference_classes_on_slab_ranks(filt):
    """The reference's LinearElasticity / Filter / PDEFilt classes on a synthetic design, 1 against 2 and 4 ranks:
    objective, constraint, sensitivity sums and ||U|| to 1e-10."""
    run = os.path.join(ROOT, "host", "slabrun")
    vals = {}
    for n in (1, 2, 4):
        r = subprocess.run([run, "-n", str(n), "--same-device", BIN, "32", "16", "16", str(filt), "-nlvls", "3"], capture_output=True,
                           text=True, timeout=300, env=dict(os.environ, PETSC_OPTIONS=" ".join(OPTS)))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        m = re.search(r"REF_ON_SHIM fx (\S+) gx (\S+) sum_dfdx (\S+) sum_dgdx (\S+) sum_xphys (\S+) normU (\S+)", r.stdout)
        assert m, r.stdout[-2000:]
        vals[n] = [float(v) for v in m.groups()]
    for n in (2, 4):
        assert vals[n] == pytest.approx(vals[1], rel=1e-10)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(TOPOPT_REF), reason="host/_refbuild/topopt_ref not built (build container only)")
@pytest.mark.parametrize("filt", [0, 2])
def test_whole_reference_program_other_filters_on_two_ranks(tmp_path, filt):
    """-filter 0 (sensitivity filter) and -filter 2 (Helmholtz PDE filter: the reference's PDEFilt with its own KSP and
    the MatCreateAIJ'ed element-to-node matrix) through the unchanged program on 2 slab processes = on one."""
    run = os.path.join(ROOT, "host", "slabrun")
    hist = {}
    for n in (1, 2):
        wd = tmp_path / ("r%d" % n)
        wd.mkdir()
        r = subprocess.run([run, "-n", str(n), "--same-device", TOPOPT_REF, "-nx", "65", "-ny", "33", "-nz", "33", "-filter", str(filt),
                            "-maxItr", "3"] + OPTS, capture_output=True, text=True, timeout=600, cwd=str(wd))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        lines = [re.sub(r"time: .*", "", ln) for ln in r.stdout.splitlines() if ln.startswith(("It.:", "State solver"))]
        assert len(lines) == 6, r.stdout[-3000:]
        hist[n] = [[float(v) for v in re.findall(r"-?\d+\.?\d*(?:[eE][+-]?\d+)?", ln)] for ln in lines]
    for a, b in zip(hist[2], hist[1]):
        assert a == pytest.approx(b, rel=2e-5, abs=2e-6), (a, b)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(BIN), reason="host/_refbuild/ref_on_shim not built (build container only)")
def test_capture_is_verified_against_what_the_reference_assembled():
    """TP_SHIM_VERIFY=1: MatSetValuesLocal normally keeps one number per element (the modulus) and one per filter (the
    radius).  In verify mode every 24x24 block of AssembleStiffnessMatrix is compared entry by entry (and index by
    index) with modulus x the first block, and ALL entries of the filter matrix that Filter::SetUp inserted
    (Filter.cc:417-433, the reference's own distances and weights) are kept and H x is compared with the device
    filter: the operators behind the names ARE the matrices the reference assembled."""
    r = subprocess.run([BIN, "32", "16", "16", "1", "-nlvls", "3"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PETSC_OPTIONS=" ".join(OPTS), TP_SHIM_VERIFY="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    m = re.search(r"verified the cone filter against the (\d+) inserted entries: max \|H x - device\| / max \|H x\| = (\S+)", r.stdout)
    assert m and int(m.group(1)) > 500000 and float(m.group(2)) < 1e-14, r.stdout[-1500:]
    assert "verified 8192 element blocks (576 entries, 24 indices each)" in r.stdout
    assert "REF_ON_SHIM fx" in r.stdout


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(TOPOPT_REF), reason="host/_refbuild/topopt_ref not built (build container only)")
@pytest.mark.parametrize("n", [1, 2])
def test_restart_through_the_reference_code(tmp_path, n):
    """The reference's own restart path (TopOpt.cc:386-512: VecLoad of x, xPhys, xo1, xo2, U, L through the compat
    binary viewer; LinearElasticity's -restartFileVecSol) on one and on two slab processes: 3 iterations, restart, 2
    more = 5 iterations straight (objective and constraint of iterations 4 and 5; the reference does not restore xold,
    so the printed design change of the first restarted iteration differs -- with real PETSc too)."""
    run = os.path.join(ROOT, "host", "slabrun")
    base = [run, "-n", str(n), "--same-device", TOPOPT_REF, "-nx", "33", "-ny", "17", "-nz", "17", "-nlvls", "3"]

    def hist(out):
        return {int(m[0]): (float(m[1]), float(m[2])) for m in re.findall(r"It\.: (\d+), True fx: (\S+), Scaled fx: \S+ gx\[0\]: (\S+),", out)}

    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir()
    b.mkdir()
    r = subprocess.run(base + ["-maxItr", "5"] + OPTS, capture_output=True, text=True, timeout=300, cwd=str(a))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    straight = hist(r.stdout)
    r = subprocess.run(base + ["-maxItr", "3"] + OPTS, capture_output=True, text=True, timeout=300, cwd=str(b))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    files = sorted((f for f in os.listdir(str(b)) if re.fullmatch(r"Restart0\d\.dat", f)), key=lambda f: os.path.getmtime(str(b / f)))
    vec = files[-1]
    r = subprocess.run(base + ["-maxItr", "5", "-restart", "1", "-restartFileVec", vec, "-restartFileItr", vec[:-4] + "_itr_f0.dat",
                               "-restartFileVecSol", "RestartSol" + vec[len("Restart"):]] + OPTS,
                       capture_output=True, text=True, timeout=300, cwd=str(b))
    assert r.returncode == 0 and "Successful restart" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    again = hist(r.stdout)
    assert sorted(again) == [4, 5] and sorted(straight) == [1, 2, 3, 4, 5]
    for it in (4, 5):
        assert again[it] == pytest.approx(straight[it], rel=2e-6, abs=2e-6)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(TOPOPT_REF), reason="host/_refbuild/topopt_ref not built (build container only)")
def test_restart_files_cross_the_boundary_both_ways(tmp_path):
    """Restart files written by the REFERENCE's code (TopOpt::WriteRestartFiles + the compat binary viewer; the iteration
    file by the reference's own ofstream) are read by the product's Python driver, and the driver's files are read by
    the reference's code (TopOpt.cc:386-512): iterations 4 and 5 equal the uninterrupted run either way."""
    import topopt_in_petsc_amd as tp
    ex, ey, ez, nlv = 32, 16, 16, 3
    h = 1.0 / ey
    kw = dict(nxyz=(ex + 1, ey + 1, ez + 1), xc=(0, ex * h, 0, 1, 0, ez * h), nlvls=nlv, rmin=2.56 * h, filter=1,
              solver=tp.SolverOptions(nlvls=nlv))
    straight = tp.TopOpt(**kw)
    want = [straight.step() for _ in range(5)][3:]
    base = [TOPOPT_REF, "-nx", str(ex + 1), "-ny", str(ey + 1), "-nz", str(ez + 1), "-nlvls", str(nlv), "-rmin", repr(2.56 * h)]

    def latest(d):
        fs = sorted((f for f in os.listdir(d) if re.fullmatch(r"Restart0\d\.dat", f)), key=lambda f: os.path.getmtime(os.path.join(d, f)))
        v = fs[-1]
        return os.path.join(d, v), os.path.join(d, v[:-4] + "_itr_f0.dat"), os.path.join(d, "RestartSol" + v[len("Restart"):])

    # ---- reference writes, product reads
    a = str(tmp_path / "a")
    os.makedirs(a)
    r = subprocess.run(base + ["-maxItr", "3"] + OPTS, capture_output=True, text=True, timeout=300, cwd=a)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    vec, itr, sol = latest(a)
    cont = tp.TopOpt(restartFileVec=vec, restartFileItr=itr, restartFileVecSol=sol, **kw)
    assert cont.itr == 3
    for w in want:
        g = cont.step()
        assert g["ksp_its"] == w["ksp_its"]
        assert g["fx"] == pytest.approx(w["fx"], rel=1e-7) and g["gx"] == pytest.approx(w["gx"], abs=1e-9)
    # ---- product writes, reference reads
    b = str(tmp_path / "b")
    first = tp.TopOpt(workdir=b, output=False, **kw)
    for _ in range(3):
        first.step()
    first.WriteRestartFiles()
    vec, itr, sol = latest(b)
    r = subprocess.run(base + ["-maxItr", "5", "-restart", "1", "-restartFileVec", vec, "-restartFileItr", itr, "-restartFileVecSol", sol] + OPTS,
                       capture_output=True, text=True, timeout=300, cwd=b)
    assert r.returncode == 0 and "Successful restart" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    got = {int(m[0]): (float(m[1]), float(m[2])) for m in re.findall(r"It\.: (\d+), True fx: (\S+), Scaled fx: \S+ gx\[0\]: (\S+),", r.stdout)}
    assert sorted(got) == [4, 5]
    for it, w in zip((4, 5), want):
        assert got[it][0] == pytest.approx(w["fx"], rel=2e-6, abs=2e-6) and got[it][1] == pytest.approx(w["gx"], abs=2e-6)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(TOPOPT_REF), reason="host/_refbuild/topopt_ref not built (build container only)")
def test_result_container_of_the_reference_writer_equals_the_products(tmp_path):
    """output_00000.dat written by the REFERENCE's MPIIO.cc (through the compat MPI-IO) against the one the product's
    Python writer (mpiio.MPIIO) produces for the same run: the same header, mesh (points, connectivity, offsets, types,
    exactly) and dump schedule, the fields (float32) equal to rounding."""
    import topopt_in_petsc_amd as tp
    from topopt_in_petsc_amd.mpiio import read_output
    ex, ey, ez, nlv = 32, 16, 16, 3
    h = 1.0 / ey
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    os.makedirs(a)
    r = subprocess.run([TOPOPT_REF, "-nx", str(ex + 1), "-ny", str(ey + 1), "-nz", str(ez + 1), "-nlvls", str(nlv), "-rmin", repr(2.56 * h),
                        "-maxItr", "3"] + OPTS, capture_output=True, text=True, timeout=300, cwd=a)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    opt = tp.TopOpt(nxyz=(ex + 1, ey + 1, ez + 1), xc=(0, ex * h, 0, 1, 0, ez * h), nlvls=nlv, rmin=2.56 * h, filter=1,
                    solver=tp.SolverOptions(nlvls=nlv), workdir=b)
    opt.run(3)
    fa, fb = read_output(os.path.join(a, "output_00000.dat")), read_output(os.path.join(b, "output_00000.dat"))
    assert fa["info"] == fb["info"] and fa["pnames"] == fb["pnames"] and fa["cnames"] == fb["cnames"]
    for k in ("points", "conn", "offsets", "types"):
        assert np.array_equal(fa[k], fb[k]), k
    assert [d[0] for d in fa["dumps"]] == [d[0] for d in fb["dumps"]] == [1, 2, 3, 4]
    for (_, pa, ca), (_, pb, cb) in zip(fa["dumps"], fb["dumps"]):
        assert np.abs(pa - pb).max() <= 1e-6 * np.abs(pa).max()
        assert np.abs(ca - cb).max() <= 1e-6


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(BIN), reason="host/_refbuild/ref_on_shim not built (build container only)")
@pytest.mark.parametrize("n,filt", [(1, 1), (2, 1), (1, 2), (2, 2)])
def test_reference_loops_elementwise_against_the_device_kernels(tmp_path, n, filt):
    """What the REFERENCE's own host loops produce for a synthetic design -- the filtered density, the compliance
    sensitivities of LinearElasticity.cc:299-437 after the chain rule of Filter.cc:120-204, the volume sensitivities and
    the state -- ELEMENT BY ELEMENT against the product's device kernels (k_objective, the filter kernels) on the same
    input; the reference side on one and on two slab processes."""
    import topopt_in_petsc_amd as tp
    ex, ey, ez, nlv = 32, 16, 16, 3
    h = 1.0 / ey
    dump = str(tmp_path / "ref.bin")
    r = subprocess.run([os.path.join(ROOT, "host", "slabrun"), "-n", str(n), "--same-device", BIN, str(ex), str(ey), str(ez), str(filt), "-nlvls", str(nlv)],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, PETSC_OPTIONS=" ".join(OPTS), REF_ON_SHIM_DUMP=dump))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    xp_r, df_r, dg_r, U_r, N_r, RHS_r = _petsc_vecs(dump)
    grid = tp.Grid(ex + 1, ey + 1, ez + 1, h)
    le = tp.LinearElasticity(grid, tp.SolverOptions(nlvls=nlv))
    le.SetUpLoadAndBC()
    flt = tp.Filter(grid, filt, 2.56 * h)   # (type 2: the reference's PDEFilt defaults = the library's)
    x = grid.synth_density()
    xt, xp, df, dg = grid.elem_vec(), grid.elem_vec(), grid.elem_vec(), grid.elem_vec()
    flt.FilterProject(x, xt, xp)
    le.ComputeObjectiveConstraintsSensitivities(df, dg, xp, 1e-9, 1.0, 3.0, 0.12)
    flt.Gradients(x, xt, df, [dg])
    # the Dirichlet and load vectors the reference's SetUpLoadAndBC wrote (the latter after :541 masked it) against
    # k_cantilever's: the same bits
    assert np.array_equal(le.N.cpu().numpy(), N_r)
    assert np.array_equal((le.RHS * le.N).cpu().numpy(), RHS_r)
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    tight = filt == 1   # the Helmholtz filter has an iterative solve (rtol 1e-8) inside
    assert rel(xp.cpu().numpy(), xp_r) <= (1e-13 if tight else 1e-7)
    assert rel(le.U.cpu().numpy(), U_r) <= (1e-8 if tight else 1e-6)
    assert rel(df.cpu().numpy(), df_r) <= (1e-8 if tight else 1e-6)
    assert rel(dg.cpu().numpy(), dg_r) <= (1e-12 if tight else 1e-6)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(TOPOPT_REF), reason="host/_refbuild/topopt_ref not built (build container only)")
def test_whole_reference_program_with_heaviside_projection(tmp_path):
    """-projectionFilter 1: the reference's Heaviside projection, its chain rule, the measure of non-discreteness and
    the beta continuation (Filter.cc:206-288) through the unchanged program against the product's driver with the
    device kernels, 12 iterations (the continuation raises beta at iteration 10)."""
    import topopt_in_petsc_amd as tp
    nit = 12
    r = subprocess.run([TOPOPT_REF, "-nx", "33", "-ny", "17", "-nz", "17", "-nlvls", "3", "-maxItr", str(nit), "-projectionFilter", "1"] + OPTS,
                       capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    ref = [[float(v) for v in m] for m in re.findall(
        r"It\.: \d+, True fx: (\S+), Scaled fx: (\S+), gx\[0\]: (\S+), ch\.: (\S+), mnd\.: (\S+),", r.stdout)]
    assert len(ref) == nit
    ex, ey, ez, nlv = 32, 16, 16, 3
    h = 1.0 / ey
    opt = tp.TopOpt(nxyz=(ex + 1, ey + 1, ez + 1), xc=(0, ex * h, 0, 1, 0, ez * h), nlvls=nlv, rmin=0.08, filter=1,
                    projectionFilter=True, solver=tp.SolverOptions(nlvls=nlv))
    for it in range(nit):
        rec = opt.step()
        fx, sfx, gx, ch, mnd = ref[it]
        assert fx == pytest.approx(rec["fx"], rel=5e-6, abs=5e-6), it
        assert gx == pytest.approx(rec["gx"], abs=5e-6) and ch == pytest.approx(rec["ch"], abs=5e-6)
        assert mnd == pytest.approx(rec["mnd"], abs=5e-6)

"""The bench contract: `python bench.py` prints exactly ONE line on stdout, a JSON object with the driver's keys plus the
`roofline` and `cpu_baseline` objects."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TP_BENCH_MEASURE_S", "0.05")   # the bench's micro-measurements: short in the tests (inherited by its subprocesses)


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "2", "--warmup", "1", "--no-cube256",
                        "--cpu-sample", "16x8x8"], capture_output=True, text=True, timeout=150, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.split("\n") if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[k], t), (k, d.get(k))
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["scaling"] in ("weak", "strong") and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] == pytest.approx(d["config"]["n_dof"] / (d["ms_per_step"] * 1e-3), rel=1e-6)
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"]) and 0 < rf["frac"] < 1
    assert "traffic" in rf and "traffic_source" in rf
    # the roofline kernel is timed where it runs (an event pair per launch inside design iterations); the back-to-back
    # figure (warm Infinity Cache) stands beside it
    assert "inside two design iterations" in rf["avg_launch_how"] and rf["avg_launch_ms"] > 0
    assert 0 < rf["back_to_back"]["frac"] < 1 and rf["back_to_back"]["avg_launch_ms"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and isinstance(cb["sample"], str)
    assert cb["cores"] == cb["omp_num_threads"] and cb["host"]["usable_cpus"] >= 1 and cb["same_mesh"] is True
    # the CPU baseline solved the line's own mesh with the line's own cycle: the GPU step is checked against it in the line
    p = d["parity"]
    assert p["ok"] is True and p["breaches"] == [], p
    assert p["its_equal"] and p["its_gpu"] == cb["cg_its"], p
    assert p["gx_abs_err"] <= 1e-13, p
    # the bounds bench.py itself enforces (exit code 4 on a breach)
    assert p["bounds"] == {"vs_arbiter_on_own_operator": 1e-10, "element_matrix": 1e-15, "vs_oracle": 1e-10, "dense_check": 1e-9, "gx_abs": 1e-13, "behind_pde_filter": 1e-6}
    # (2) the operator the kernels apply: the library's export is the restatement the arbiter ran on, and it is KE to 1e-15
    em = p["element_matrix"]
    assert em["library_export_equals_restatement"] is True and 0 < em["KE_eff_vs_KE"] <= 1e-15, em
    # round 6: the packed form keeps KE's translation residues (three values) -- its translation energy is KE's, to one rounding
    assert em["translation_residue_kept"] is True and 0 < em["translation_energy_KE"] and em["translation_energy_KE_eff_vs_KE"] <= 1e-3, em
    # ... and the Krylov operator (A p, initial residual, MatMult) also KE's whole answer to a rigid translation
    assert 0 < em["KE_krylov_vs_KE"] <= 1e-15 and em["translation_column_defect_KE_krylov"] <= 1e-18 < em["translation_column_defect_KE_eff"], em
    # (1) the GPU against the 80-bit arbiter on that operator: iteration counts, ||r_k||, compliance -- 1e-10, as north_star has it
    ar = p["arbiter"]
    ge = ar["gpu_vs_arbiter_on_KE_eff"]
    assert ge["its_equal"] and ge["fx_rel_err"] <= 1e-10 and ge["hist_max_rel_err"] <= 1e-10, ar
    assert ar["oracle_vs_arbiter_on_KE"]["its_equal"] and ar["oracle_vs_arbiter_on_KE"]["hist_max_rel_err"] <= 1e-10
    # (3) on this small mesh the conditioning is mild: the GPU is within 1e-10 of the oracle on KE as well
    assert p["fx_rel_err"] <= 1e-10 and p["hist_max_rel_err_all"] <= 1e-10, p
    # the converged step (rtol 1e-12): compliance and raw sensitivities are solver independent
    c = p["converged"]
    assert c["rtol"] == 1e-12 and c["rel_residual_gpu"] <= 1e-12 and c["its_gpu"] == c["its_cpu"] == c["its_arbiter"] == c["its_arbiter_on_KE_eff"], c
    for who in ("gpu_vs_arbiter_on_KE_eff", "gpu_vs_oracle"):
        assert c[who]["fx_rel_err"] <= 1e-10 and c[who]["dfdx_max_err_rel_to_max"] <= 1e-10, (who, c)


@pytest.mark.gpu
def test_bench_exits_nonzero_when_a_parity_bound_breaks():
    """a bound of the parity object broken (here: made unreachable through TP_BENCH_TEST_PARITY_BOUND) -> the line is still
    printed, with "ok": false and the breach named, and the exit code is 4"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "1", "--warmup", "1", "--no-cube256",
                        "--no-stated-cycle", "--cpu-sample", "16x8x8"], capture_output=True, text=True, timeout=150, cwd=ROOT,
                       env=dict(os.environ, TP_BENCH_TEST_PARITY_BOUND="1e-30"))
    assert r.returncode == 4, (r.returncode, r.stderr[-2000:])
    lines = [ln for ln in r.stdout.split("\n") if ln.strip()]
    assert len(lines) == 1
    p = json.loads(lines[0])["parity"]
    assert p["ok"] is False and "converged.gpu_vs_arbiter_on_KE_eff.fx_rel_err" in p["breaches"] and "parity bounds broken" in r.stderr


@pytest.mark.gpu
def test_design_loop_mode_against_the_oracle_loop():
    """`bench.py --design-loop N` (real iterates: main.cc:54-123 from the uniform start, warm-started solves LinearElasticity.cc:647,
    device MMA) at 64 x 32 x 32 with the metric mesh's recipe (4 levels, Chebyshev(2), level 2 cycled three times, exact coarse solve):
    the first 5 iterations' CG iteration counts, fx, gx and design change against the same loop driven by the oracle (assembled CSR,
    oracle MMA)."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cant64", "--steps", "1", "--warmup", "1", "--no-cube256",
                        "--no-stated-cycle", "--no-cpu-baseline", "--design-loop", "5", "--design-loop-records"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.split("\n") if ln.strip()][0])
    dl = d["config"]["design_loop"]
    assert dl["iterations"] == 5 and len(dl["records"]) == 5 and dl["first"]["iterations"] == "1-5" and dl["giveups_xcdoff_deferoff"] == [0, False, False]
    ex, ey, ez, nlv = 64, 32, 32, 4
    nx, ny, nz, h = ex + 1, ey + 1, ez + 1, 1.0 / ey
    rmin = 2.56 * h
    KE = orc.hex8_ke_box(h, h, h, 0.3)
    N, R = orc.cantilever_bc(nx, ny, nz, h)
    flt = orc.Filter(nx, ny, nz, h, rmin)
    mg = orc.MG(nx, ny, nz, 3, nlv, 2, 20)
    mg.set_coarse_direct(True)
    mg.set_cycles([1, 3, 1])
    x = np.full(ex * ey * ez, 0.12)
    xt, xp = flt.project(1, x)
    mma = orc.MMA(x, 1)
    xold = x.copy()
    U = np.zeros(3 * nx * ny * nz)
    fscale = None
    for it in range(5):
        rec = dl["records"][it]
        mg.assemble(KE, orc.simp(xp), N)
        U, its, hist = mg.solve(R * N, x0=U, rtol=1e-5)
        fx, gx, df, dg = orc.compliance_sens(nx, ny, nz, KE, U, xp)
        if fscale is None:
            fscale = 10.0 / fx
        df = flt.gradient(1, x, xt, df * fscale)
        dg = flt.gradient(1, x, xt, dg)
        xmin, xmax = mma.SetOuterMovelimit(0.0, 1.0, 0.2, x)
        x = mma.Update(x, df, [gx], [dg], xmin, xmax)
        ch = mma.DesignChange(x, xold)
        xt, xp = flt.project(1, x)
        assert rec["itr"] == it + 1 and rec["cg_its"] == its, (it, rec, its)
        assert rec["fx"] == pytest.approx(fx, rel=1e-8), (it, rec["fx"], fx)
        assert rec["gx"] == pytest.approx(gx, abs=1e-10) and rec["ch"] == pytest.approx(ch, abs=1e-7)
        assert rec["mnd"] == pytest.approx(orc.mnd(xp), rel=1e-7)


def _check_two_rank_line(r, scaling):
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.split("\n") if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert "error" not in d, d
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["value"] > 0
    assert d["config"]["parallelism"] == "zslab2" and d["config"]["halo_overlap"] > 0 and d["config"]["comm_ranks"] == 2
    ez = 32 if scaling == "weak" else 16
    assert "32x16x%d elements" % ez in d["config"]["workload"]
    assert "cpu_baseline" not in d          # rank 0 of a 1-GPU job only
    cr = d["config"]["comm_report"]         # (gloo + host staging here: no RCCL communicator to report ranks)
    assert cr["ranks"] == 2 and cr["rccl_ranks_seen"] == 0 and "hooks" in cr["path"]
    # the complementary reading of the metric is timed in the same run
    o = d["other_scaling"]
    assert o["scaling"] == ("strong" if scaling == "weak" else "weak") and o["value"] > 0 and o["cg_its"] > 0
    assert "32x16x%d elements" % (16 if scaling == "weak" else 32) in o["mesh"]
    # round 6: where the communication time goes (two extra steps with the timer on), and the slab run against the same global
    # mesh solved on ONE GPU without slabs: the same iteration count, every ||r_k|| and fx to 1e-10 (rank-ordered sums of the
    # host-staged hooks; with RCCL the order of the sum is the library's own)
    ct = d["config"]["comm_time"]["per_step"]
    assert set(ct) == {"halo_blocking", "halo_overlapped", "all_reduce", "all_gather"}
    assert ct["all_reduce"]["calls"] > 0 and ct["all_reduce"]["host_ms"] > 0 and (ct["halo_blocking"]["calls"] + ct["halo_overlapped"]["calls"]) > 0
    one = d["config"]["slabs_vs_one_gpu"]
    assert one["its_equal"] and one["hist_max_rel_err"] <= 1e-10 and one["fx_rel_err"] <= 1e-10, one


TWO_RANKS = ["--gpus", "2", "--same-device", "--backend", "gloo", "--workload", "tiny", "--steps", "2", "--warmup", "1"]


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_spawns_its_own_ranks(scaling):
    """`python bench.py --gpus 2` without a launcher starts its two ranks itself (bench.py: spawn_ranks -- one process
    group per rank, a deadline, no rank left behind); here both slabs share the one GPU of the box (--same-device,
    host-staged gloo hooks).  The whole run takes seconds: the limit is what guards the suite against a hang."""
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + TWO_RANKS + ["--scaling", scaling, "--budget-s", "200"],
                       capture_output=True, text=True, timeout=240, cwd=ROOT)
    _check_two_rank_line(r, scaling)
    assert time.time() - t0 < 230


@pytest.mark.gpu
def test_bench_ranks_make_the_same_calls_whatever_their_clocks():
    """Round 3's hang: the micro-measurement helper warmed up for a fixed TIME, so the number of Chebyshev sweeps -- each with
    halo exchanges -- depended on a rank's own clock; one batch of difference and one rank waits for ever.  Here rank 1 is
    slowed down inside that helper (TP_BENCH_TEST_SLOW_RANK): the run must still end, with its line."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + TWO_RANKS + ["--no-other-scaling", "--budget-s", "120"],
                       capture_output=True, text=True, timeout=150, cwd=ROOT, env=dict(os.environ, TP_BENCH_TEST_SLOW_RANK="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.split("\n") if ln.strip()][-1])
    assert "error" not in d and d["n_gpus"] == 2 and d["value"] > 0


@pytest.mark.gpu
def test_bench_under_torch_distributed_run():
    """the driver's form for N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py")] + TWO_RANKS,
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    _check_two_rank_line(r, "weak")


@pytest.mark.gpu
def test_bench_ends_with_an_error_line_when_a_phase_overruns():
    """the wall-clock watchdog: a phase over its limit -> stacks on stderr, ONE line with an "error" key, exit code 3"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "2", "--warmup", "1", "--no-cube256",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=150, cwd=ROOT, env=dict(os.environ, TP_BENCH_TEST_OVERRUN="warm-up"))
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    lines = [ln for ln in r.stdout.split("\n") if ln.strip()]
    assert len(lines) == 1 and "warm-up" in json.loads(lines[0])["error"] and "exceeded its wall-clock limit" in r.stderr


def test_spawner_kills_ranks_that_never_finish():
    """CPU: bench.py's own launcher comes back when its ranks hang (here: two sleepers), kills their process groups and
    reports it in the contract's shape"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--budget-s", "3"], capture_output=True, text=True,
                       timeout=60, cwd=ROOT, env=dict(os.environ, TP_BENCH_TEST_SLEEPER="1"))
    assert r.returncode == 4, (r.returncode, r.stderr[-1000:])
    d = json.loads([ln for ln in r.stdout.split("\n") if ln.strip()][-1])
    assert d["value"] is None and "did not finish" in d["error"] and d["n_gpus"] == 2
    assert "killed" in r.stderr


def test_bench_workloads_are_consistent():
    """CPU check of bench.py's workload table: every mesh coarsens nlvls - 1 times, a cycle pattern has one entry per
    level that has a coarser one, and the slab geometries of the driver's scaling runs (1, 2, 4, 8 GPUs, weak and --
    where the layers divide -- strong) pass the library's own rule for slabs (topopt_amd.hip: every distributed level
    keeps two element layers per rank; with three or more levels the coarsest one is the replicated copy)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert {"cantilever128", "c1", "c2", "c3", "c4", "c5", "cube256"} <= set(b.WORKLOADS)
    for name, w in b.WORKLOADS.items():
        ex, ey, ez = w["el"]
        f = 1 << (w["nlvls"] - 1)
        assert ex % f == 0 and ey % f == 0 and ez % f == 0, name
        assert 1 <= w["nsmooth"] <= 8 and 1 <= w["ncoarse"] <= 96, name     # RUN_MAXK of csrc/coarse_run.h
        if "cycles" in w:
            c = [int(v) for v in w["cycles"].split(",")]
            assert len(c) == w["nlvls"] - 1 and all(1 <= v <= 4 for v in c) and c[-1] == 1 and c[0] == 1, name
        last_distributed = w["nlvls"] - 2 if w["nlvls"] >= 3 else w["nlvls"] - 1
        for world in (2, 4, 8):
            assert (ez >> last_distributed) >= 2, (name, "weak", world)          # weak: ez layers per rank
            if ez % world == 0 and (ez // world) % f == 0:                        # strong: bench.py refuses otherwise
                assert ((ez // world) >> last_distributed) >= 2 or name in ("c1",), (name, "strong", world)
    assert b.WORKLOADS["c2"]["nlvls"] == 3 and b.WORKLOADS["c5"]["nlvls"] == 4      # stated by BASELINE.json configs[1], [4]

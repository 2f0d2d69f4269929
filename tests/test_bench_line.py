"""The bench contract: `python bench.py` prints exactly ONE line on stdout, a JSON object with the driver's keys plus the
`roofline` and `cpu_baseline` objects."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "2", "--warmup", "1", "--no-cube256",
                        "--cpu-sample", "16x8x8"], capture_output=True, text=True, timeout=600, cwd=ROOT)
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


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_spawns_its_own_ranks(scaling):
    """`python bench.py --gpus 2` without a launcher re-executes itself under torch.distributed.run (what the driver's
    scaling run does); here both slabs share the one GPU of the box (--same-device, host-staged gloo hooks)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--backend", "gloo",
                        "--workload", "tiny", "--steps", "2", "--warmup", "1", "--scaling", scaling],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.split("\n") if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["value"] > 0
    assert d["config"]["parallelism"] == "zslab2" and d["config"]["halo_overlap"] > 0
    ez = 32 if scaling == "weak" else 16
    assert "32x16x%d elements" % ez in d["config"]["workload"]
    assert "cpu_baseline" not in d          # rank 0 of a 1-GPU job only
    # the complementary reading of the metric is timed in the same run
    o = d["other_scaling"]
    assert o["scaling"] == ("strong" if scaling == "weak" else "weak") and o["value"] > 0 and o["cg_its"] > 0
    assert "32x16x%d elements" % (16 if scaling == "weak" else 32) in o["mesh"]


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

"""The C++ host mirror (host/topopt_host.h + host/main.cc): builds with plain g++ against the C ABI
and, on a GPU box, reproduces the Python driver's iteration history."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    from topopt_in_petsc_amd import lib
    lib.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host")], stdout=subprocess.DEVNULL)
    return os.path.join(ROOT, "host", "topopt")


def test_cpp_host_builds():
    assert os.path.exists(_build())


@pytest.mark.gpu
def test_cpp_driver_matches_python_driver():
    exe = _build()
    out = subprocess.run([exe, "-nx", "33", "-ny", "17", "-nz", "17", "-nlvls", "3", "-maxItr", "5", "-rmin", "0.16"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    fx_cpp = [float(v) for v in re.findall(r"True fx: ([0-9.eE+-]+)", out.stdout)]
    its_cpp = [int(v) for v in re.findall(r"State solver:  iter: (\d+)", out.stdout)]
    assert len(fx_cpp) == 5
    import topopt_in_petsc_amd as tp
    opt = tp.TopOpt(nxyz=(33, 17, 17), nlvls=3, rmin=0.16)
    hist = [opt.step() for _ in range(5)]
    assert its_cpp == [h["ksp_its"] for h in hist]
    for a, h in zip(fx_cpp, hist):
        assert a == pytest.approx(h["fx"], rel=1e-5)   # printed with 6 decimals
    assert "# final volume fraction 0.1" in out.stdout

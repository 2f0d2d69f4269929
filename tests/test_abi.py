"""CPU-side checks of the drop-in boundary: the library loads and exports every
symbol include/topopt_amd.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "topopt_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tp_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from topopt_in_petsc_amd import lib
    assert sorted(lib.SYMBOLS) == _declared()


def test_library_exports_every_declared_symbol():
    from topopt_in_petsc_amd import lib
    so = lib.build()
    dll = ctypes.CDLL(so)
    for name in _declared():
        assert hasattr(dll, name), "libtopopt_amd.so lacks %s" % name
    assert lib.load_library() is not None


def test_default_options_match_reference_defaults():
    from topopt_in_petsc_amd import lib, SolverOptions
    o = lib.SolverOpts()
    lib.load_library().tp_solver_default_opts(ctypes.byref(o))
    # LinearElasticity.cc:22-23, :621-635
    assert (o.nlvls, o.nu, o.rtol, o.atol, o.dtol, o.max_it, o.nsmooth, o.ncoarse) == (4, 0.3, 1e-5, 1e-50, 1e5, 200, 4, 30)
    d = SolverOptions()
    assert (d.nlvls, d.nu, d.rtol, d.max_it, d.nsmooth, d.ncoarse, d.cheb_lo, d.cheb_hi) == \
        (o.nlvls, o.nu, o.rtol, o.max_it, o.nsmooth, o.ncoarse, o.cheb_lo, o.cheb_hi)


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import topopt_in_petsc_amd as tp
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tp.Grid(9, 5, 5, 0.25)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "topopt_in_petsc_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("no CPU fallback", "").lower() or f == "__init__.py" and False, \
                    "%s mentions the oracle" % f

"""The arbiter: the oracle's own algorithm (oracle/topopt_oracle.c, the same source) executed in x87 extended precision.

TEST INFRASTRUCTURE ONLY (tests/, bench.py's cpu_baseline leg) -- like oracle.py.

Why it exists ("next" 1): at the metric's own 128^3 mesh the HIP path and the oracle agree on the CG
residual history to 2.6e-10 and on the compliance to 1.6e-10 at rtol 1e-5 -- both double-precision results of the same
algorithm, summed in different orders (assembled CSR rows against the kernels' block form).  Neither is "the" answer.
The arbiter runs the SAME algorithm on the SAME double-precision inputs (KE, moduli, Dirichlet vector, load) with every
operation in 80-bit `long double` (unit round-off 5.4e-20 instead of 1.1e-16), so its trajectory is the exact-arithmetic
one to ~1e-13 at these sizes; `|gpu - arbiter|` against `|oracle - arbiter|` says which side of a disagreement is off,
and their common size is the rounding sensitivity of the problem itself.

This module is oracle.py executed a second time with REAL = numpy.longdouble against liboracle_ld.so (built by
oracle/Makefile from the same C file by `sed s/double/long double/`); every class and function of oracle.py is
available here with the same signature (MMA excepted), taking and returning numpy.longdouble arrays.
"""
import os as _os

_FLAVOUR = "ld"
_src = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "oracle.py")
with open(_src) as _f:
    exec(compile(_f.read(), _src, "exec"), globals())

"""MI355X-native hot path of TopOpt_in_PETSc (assembly + CG/GMG solve + filters).

The compute lives in libtopopt_amd.so (hand-written HIP for gfx950) behind the
C ABI of include/topopt_amd.h.  This package is the thin host-side mirror used
by the tests and the benchmark: PyTorch only supplies device memory, the stream
and torch.distributed for the z-slab halo exchange.
"""
from .lib import load_library, LibraryMissing  # noqa: F401
from .partition import SlabPartition  # noqa: F401
from .api import Grid, LinearElasticity, Filter, MMA, SolverOptions, TopOptError  # noqa: F401
from .driver import TopOpt  # noqa: F401

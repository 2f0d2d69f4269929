"""ctypes loader of libtopopt_amd.so.  There is NO fallback: if the HIP library
is missing the product refuses to run (tests on a GPU box must exercise the
native code, never a substitute)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("TP_LIB") or os.path.join(_HERE, "libtopopt_amd.so")  # TP_LIB: experiment builds
_LIB = None


class LibraryMissing(RuntimeError):
    pass


def build(force=False):
    """hipcc --offload-arch=gfx950 build of the library (cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    deps = [os.path.join(src, f) for f in os.listdir(src)] + [os.path.join(_HERE, "..", "include", "topopt_amd.h")]
    if force or not os.path.exists(SO_PATH) or any(os.path.getmtime(d) > os.path.getmtime(SO_PATH) for d in deps):
        subprocess.check_call(["make", "-C", src], stdout=subprocess.DEVNULL)
    return SO_PATH


class GridOpts(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("hx", C.c_double), ("hy", C.c_double),
                ("hz", C.c_double), ("rank", C.c_int), ("nranks", C.c_int), ("device", C.c_int),
                ("stream", C.c_void_p), ("comm", C.c_void_p)]


class SolverOpts(C.Structure):
    _fields_ = [("nlvls", C.c_int), ("nu", C.c_double), ("rtol", C.c_double), ("atol", C.c_double),
                ("dtol", C.c_double), ("max_it", C.c_int), ("nsmooth", C.c_int), ("ncoarse", C.c_int),
                ("cheb_lo", C.c_double), ("cheb_hi", C.c_double), ("nlanczos", C.c_int), ("fine_eig", C.c_int),
                ("ksp_mode", C.c_int), ("restart", C.c_int), ("smooth_pc", C.c_int), ("coarse_pc", C.c_int),
                ("coarse_restart", C.c_int), ("coarse_rtol", C.c_double), ("coarse_direct", C.c_int)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_long)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
DIRECT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long)
INPLACE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)
SETSTREAM_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)


class Comm(C.Structure):
    _fields_ = [("user", C.c_void_p), ("send_lo", C.c_void_p), ("send_hi", C.c_void_p), ("recv_lo", C.c_void_p),
                ("recv_hi", C.c_void_p), ("red", C.c_void_p), ("cap", C.c_long), ("exchange", EXCHANGE_FN),
                ("allreduce_sum", ALLREDUCE_FN), ("gather", C.c_void_p), ("allgather", EXCHANGE_FN),
                ("exchange_direct", DIRECT_FN), ("allreduce_inplace", INPLACE_FN), ("set_stream", SETSTREAM_FN)]


# every symbol include/topopt_amd.h declares: (restype, argtypes)
_vp, _i, _d, _l = C.c_void_p, C.c_int, C.c_double, C.c_long
SYMBOLS = {
    "tp_grid_create": (_i, [C.POINTER(_vp), C.POINTER(GridOpts)]),
    "tp_grid_destroy": (_i, [_vp]),
    "tp_rccl_load": (_i, [C.c_char_p]),
    "tp_rccl_unique_id": (_i, [_vp]),
    "tp_grid_use_rccl": (_i, [_vp, _vp]),
    "tp_grid_use_rccl2": (_i, [_vp, _vp, _vp]),
    "tp_grid_comm_stats": (_i, [_vp, C.POINTER(_l), C.POINTER(_l)]),
    "tp_grid_comm_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "tp_grid_comm_timer": (_i, [_vp, _i]),
    "tp_grid_comm_timer_read": (_i, [_vp, C.POINTER(_l), C.POINTER(_d), C.POINTER(_d)]),
    "tp_grid_kernel_timer": (_i, [_vp, _i]),
    "tp_grid_kernel_timer_read": (_i, [_vp, C.POINTER(_d), C.POINTER(_l)]),
    "tp_grid_kernel_timer_read2": (_i, [_vp, C.POINTER(_d), C.POINTER(_l), C.POINTER(_d)]),
    "tp_grid_drop_rccl": (_i, [_vp]),
    "tp_grid_overlapped_halos": (_l, [_vp]),
    "tp_grid_comm_selfcheck": (_i, [_vp, C.POINTER(_i)]),
    "tp_grid_reduction_selftest": (_i, [_vp, _l, _i, C.POINTER(_i)]),
    "tp_rccl_selftest": (_i, [_i, _vp, _l, C.POINTER(_d)]),
    "tp_grid_local_nodes": (_l, [_vp]),
    "tp_grid_local_elems": (_l, [_vp]),
    "tp_grid_owned_node_offset": (_l, [_vp]),
    "tp_grid_owned_nodes": (_l, [_vp]),
    "tp_grid_node_z0": (_i, [_vp]),
    "tp_grid_halo_nodes": (_i, [_vp, _vp, C.c_int]),
    "tp_grid_elem_z0": (_i, [_vp]),
    "tp_set_device": (_i, [C.c_int]),
    "tp_malloc": (_i, [C.POINTER(_vp), C.c_size_t]),
    "tp_free": (_i, [_vp]),
    "tp_memcpy_h2d": (_i, [_vp, _vp, C.c_size_t]),
    "tp_memcpy_d2h": (_i, [_vp, _vp, C.c_size_t]),
    "tp_sync": (_i, [_vp]),
    "tp_solver_default_opts": (None, [C.POINTER(SolverOpts)]),
    "tp_abi_version": (_i, []),
    "tp_solver_opts_size": (C.c_ulong, []),
    "tp_elasticity_create": (_i, [C.POINTER(_vp), _vp, C.POINTER(SolverOpts)]),
    "tp_elasticity_create_ke": (_i, [C.POINTER(_vp), _vp, C.POINTER(SolverOpts), _vp]),
    "tp_elasticity_destroy": (_i, [_vp]),
    "tp_elasticity_get_ke": (_i, [_vp, _vp]),
    "tp_elasticity_get_ke_effective": (_i, [_vp, _vp, _vp]),
    "tp_elasticity_get_ke_krylov": (_i, [_vp, _vp, _vp]),
    "tp_elasticity_cantilever": (_i, [_vp, _vp, _vp]),
    "tp_elasticity_set_bc": (_i, [_vp, _vp]),
    "tp_elasticity_assemble": (_i, [_vp, _vp, _d, _d, _d]),
    "tp_elasticity_apply": (_i, [_vp, _vp, _vp]),
    "tp_elasticity_apply_krylov": (_i, [_vp, _vp, _vp]),
    "tp_elasticity_solve": (_i, [_vp, _vp, _vp, C.POINTER(_i), C.POINTER(_d), C.POINTER(_d), _vp, _i]),
    "tp_elasticity_objective": (_i, [_vp, _vp, _vp, _d, _d, _d, _d, C.POINTER(_d), C.POINTER(_d), _vp, _vp]),
    "tp_elasticity_objective_only": (_i, [_vp, _vp, _vp, _d, _d, _d, _d, C.POINTER(_d), C.POINTER(_d)]),
    "tp_elasticity_sensitivities": (_i, [_vp, _vp, _vp, _d, _d, _d, _vp, _vp]),
    "tp_elasticity_petsc_options": (_i, [_vp, C.c_char_p, C.c_size_t]),
    "tp_elasticity_level_count": (_i, [_vp]),
    "tp_elasticity_level_nodes": (_l, [_vp, _i]),
    "tp_elasticity_level_lambda": (_d, [_vp, _i]),
    "tp_elasticity_level_lambda_min": (_d, [_vp, _i]),
    "tp_elasticity_coarse_direct_active": (_i, [_vp]),
    "tp_xcd_status": (_i, [C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "tp_elasticity_level_apply": (_i, [_vp, _i, _vp, _vp]),
    "tp_elasticity_level_diag": (_i, [_vp, _i, _vp]),
    "tp_elasticity_set_cycles": (_i, [_vp, _vp, _i]),
    "tp_elasticity_precond": (_i, [_vp, _vp, _vp]),
    "tp_elasticity_level_pc": (_i, [_vp, _i, _i, _vp, _vp]),
    "tp_elasticity_level_gmres": (_i, [_vp, _i, _i, _i, _i, _d, _vp, _vp, _i, C.POINTER(_i)]),
    "tp_elasticity_smooth": (_i, [_vp, _i, _vp, _vp, _i, _i]),
    "tp_elasticity_restrict": (_i, [_vp, _i, _vp, _vp]),
    "tp_elasticity_prolong_add": (_i, [_vp, _i, _vp, _vp]),
    "tp_elasticity_last_stats": (_i, [_vp, C.POINTER(_d), C.POINTER(_d), C.POINTER(_l)]),
    "tp_filter_create": (_i, [C.POINTER(_vp), _vp, _i, _d, C.POINTER(SolverOpts)]),
    "tp_filter_destroy": (_i, [_vp]),
    "tp_filter_stencil_width": (_i, [_vp]),
    "tp_filter_get_hs": (_i, [_vp, _vp]),
    "tp_filter_get_kf": (_i, [_vp, _vp]),
    "tp_filter_project": (_i, [_vp, _vp, _vp, _vp, _i, _d, _d]),
    "tp_filter_gradients": (_i, [_vp, _vp, _vp, _vp, _i, C.POINTER(_vp), _i, _d, _d]),
    "tp_filter_mnd": (_i, [_vp, _vp, C.POINTER(_d)]),
    "tp_filter_last_pde_its": (_i, [_vp, C.POINTER(_i), C.POINTER(_d)]),
    "tp_pdefilter_elem_to_node": (_i, [_vp, _vp, _vp]),
    "tp_pdefilter_solve": (_i, [_vp, _vp, _vp]),
    "tp_pdefilter_node_to_elem": (_i, [_vp, _vp, _vp]),
    "tp_pdefilter_apply": (_i, [_vp, _vp, _vp]),
    "tp_mma_create": (_i, [C.POINTER(_vp), _vp, _l, _l, _i, _vp]),
    "tp_mma_destroy": (_i, [_vp]),
    "tp_mma_set_outer_movelimit": (_i, [_vp, _d, _d, _d, _vp, _vp, _vp]),
    "tp_mma_update": (_i, [_vp, _vp, _vp, C.POINTER(_d), C.POINTER(_vp), _vp, _vp, C.POINTER(_i)]),
    "tp_mma_design_change": (_i, [_vp, _vp, _vp, C.POINTER(_d)]),
    "tp_mma_get_state": (_i, [_vp, C.POINTER(_d), C.POINTER(_d), C.POINTER(_i)]),
    "tp_vec_axpby": (_i, [_vp, _vp, _d, _vp, _d, _l]),
    "tp_vec_pointwise": (_i, [_vp, _vp, _vp, _vp, _i, _l]),
    "tp_vec_dot": (_i, [_vp, _vp, _vp, _l, C.POINTER(_d)]),
    "tp_elasticity_set_tolerances": (_i, [_vp, _d, _d, _d, _i]),
    "tp_filter_mult_h": (_i, [_vp, _vp, _vp]),
    "tp_mma_restart_get": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "tp_mma_restart_set": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "tp_vec_scale": (_i, [_vp, _vp, _d, _l]),
    "tp_vec_set": (_i, [_vp, _vp, _d, _l]),
    "tp_synth_density": (_i, [_vp, _vp, C.c_uint64]),
}


ABI_VERSION = 4   # TP_ABI_VERSION of include/topopt_amd.h


def load_library():
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise LibraryMissing(
                "libtopopt_amd.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                "the MI355X hot path has no CPU fallback")
        lib = C.CDLL(SO_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        # include/topopt_amd.h: the option structs grow at their end; a stale binding must not reach the library
        if lib.tp_abi_version() != ABI_VERSION or lib.tp_solver_opts_size() != C.sizeof(SolverOpts):
            raise LibraryMissing("libtopopt_amd.so (ABI %d, tp_solver_opts of %d bytes) does not match this binding (ABI %d, %d bytes): rebuild"
                                 % (lib.tp_abi_version(), lib.tp_solver_opts_size(), ABI_VERSION, C.sizeof(SolverOpts)))
        _LIB = lib
    return _LIB

"""Host-side mirror of the reference's operator interface for the hot path.

Class and method names follow the reference (LinearElasticity.h:21-109,
Filter.h:34-92) so that tests read like the reference's call sites in main.cc;
vectors are 1-D float64 torch tensors on the GPU standing in for PETSc Vecs
(local slab layout, see include/topopt_amd.h).  Everything numerical happens in
libtopopt_amd.so.
"""
import ctypes as C
import os
from dataclasses import dataclass

import torch

from . import lib as _lib
from .comm import SlabComm
from .partition import SlabPartition


class TopOptError(RuntimeError):
    def __init__(self, code, where):
        names = {1: "TP_ERR_ARG", 2: "TP_ERR_STATE", 3: "TP_ERR_DIVERGED", 4: "TP_ERR_COMM"}
        msg = names.get(code, "hipError %d" % (code - 1000) if code >= 1000 else str(code))
        super().__init__("%s failed: %s" % (where, msg))
        self.code = code


def _chk(code, where):
    if code != 0:
        raise TopOptError(code, where)


def _ptr(t):
    if t is None:
        return None
    assert t.dtype == torch.float64 and t.is_contiguous() and t.is_cuda, "need a contiguous float64 GPU tensor"
    return t.data_ptr()


@dataclass
class SolverOptions:
    """LinearElasticity.cc:621-635 defaults; smoother = Chebyshev(nsmooth)/Jacobi."""
    nlvls: int = 4
    nu: float = 0.3
    rtol: float = 1.0e-5
    atol: float = 1.0e-50
    dtol: float = 1.0e5
    max_it: int = 200
    nsmooth: int = 4
    ncoarse: int = 30
    cheb_lo: float = 0.1
    cheb_hi: float = 1.1
    nlanczos: int = 10
    fine_eig: int = 0   # 0: element bound on the fine level, 1: Lanczos estimate
    # ksp_mode 1: the configuration the reference hard-codes (FGMRES(restart) + V-cycle with GMRES(nsmooth) smoothers and
    # a GMRES(coarse_restart) coarse solve, PCSOR / PCJACOBI; LinearElasticity.cc:620-746, PDEFilter.cc:276-378) run as
    # written -- a correctness mode on one device; 0: CG + Chebyshev/Jacobi, the fast path
    ksp_mode: int = 0
    restart: int = 100
    smooth_pc: int = 1      # 0 PCJACOBI, 1 PCSOR
    coarse_pc: int = 1
    coarse_restart: int = 30
    coarse_rtol: float = 1.0e-8
    # ksp_mode 0: coarsest level solved exactly (banded Cholesky + explicit triangular inverse, csrc/coarse_direct.h) where it
    # has 449 .. 4096 rows on one rank (2: also below 449, where the Chebyshev run stays inside one workgroup); 0: Chebyshev run
    coarse_direct: int = 0

    @classmethod
    def reference_elasticity(cls, **kw):
        """LinearElasticity::SetUpSolver as hard-coded (LinearElasticity.cc:620-746)"""
        return cls(**{**dict(ksp_mode=1, restart=100, nsmooth=4, ncoarse=30, smooth_pc=1, coarse_pc=1, coarse_restart=30,
                             coarse_rtol=1e-8, rtol=1e-5, atol=1e-50, dtol=1e5, max_it=200), **kw})

    @classmethod
    def reference_pdefilter(cls, **kw):
        """PDEFilt::SetUpSolver as hard-coded (PDEFilter.cc:276-378)"""
        return cls(**{**dict(ksp_mode=1, nlvls=3, restart=20, nsmooth=1, ncoarse=10, smooth_pc=0, coarse_pc=0, coarse_restart=10,
                             coarse_rtol=1e-8, rtol=1e-8, atol=1e-50, dtol=1e3, max_it=60), **kw})

    def c_struct(self):
        return _lib.SolverOpts(self.nlvls, self.nu, self.rtol, self.atol, self.dtol, self.max_it, self.nsmooth,
                               self.ncoarse, self.cheb_lo, self.cheb_hi, self.nlanczos, self.fine_eig, self.ksp_mode,
                               self.restart, self.smooth_pc, self.coarse_pc, self.coarse_restart, self.coarse_rtol,
                               self.coarse_direct)


class Grid:
    """The DMDA stand-in: global node counts, element size, z-slab partition."""

    def __init__(self, nx, ny, nz, h, rank=0, nranks=1, device=None, group=None):
        self.L = _lib.load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the MI355X hot path has no CPU fallback")
        self.part = SlabPartition(nx, ny, nz, rank, nranks)
        self.h = (h, h, h) if isinstance(h, (int, float)) else tuple(h)
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.stream = torch.cuda.current_stream(self.device)
        self.comm = None
        comm_p = None
        if nranks > 1:
            self.comm = SlabComm(self.part, self.device, group)
            self.comm.stream = self.stream   # the hooks issue their collectives on the library's stream
            comm_p = C.cast(C.pointer(self.comm.c_struct), C.c_void_p)
        self._opts = _lib.GridOpts(nx, ny, nz, self.h[0], self.h[1], self.h[2], rank, nranks, self.device.index,
                                   self.stream.cuda_stream, comm_p)
        self.handle = C.c_void_p()
        self._children = []   # weak references to the solvers / filters / optimisers built on this grid (close() order)
        _chk(self.L.tp_grid_create(C.byref(self.handle), C.byref(self._opts)), "tp_grid_create")
        self.comm_kind = "none" if nranks == 1 else "torch.distributed hooks"
        if nranks > 1 and self.comm.backend == "nccl" and os.environ.get("TP_COMM", "rccl") != "torch":
            self._use_rccl(group)

    @property
    def halo_overlap(self):
        """number of halos that travelled on the second stream, overlapped with interior planes (0 = none)"""
        return int(self.L.tp_grid_overlapped_halos(self.handle))

    def comm_stats(self):
        """(halo exchanges, all-reduces) the in-library RCCL path has issued so far; (0, 0) on the hooks / one rank"""
        ex, red = C.c_long(0), C.c_long(0)
        self.L.tp_grid_comm_stats(self.handle, C.byref(ex), C.byref(red))
        return ex.value, red.value

    def comm_report(self):
        """what the slab exchange runs on, for the bench line: path, the ranks the RCCL communicator itself reports, whether
        the in-library path passed its collective self-check (or why the grid went back to the host-staged hooks)"""
        n, two = C.c_int(0), C.c_int(0)
        self.L.tp_grid_comm_info(self.handle, C.byref(n), C.byref(two))
        return {"path": self.comm_kind, "ranks": self.part.nranks, "rccl_ranks_seen": n.value, "two_communicators": bool(two.value),
                "selfcheck": getattr(self, "_rccl_selfcheck", "not run (no in-library RCCL path requested)" if self.part.nranks > 1 else "one rank")}

    def reduction_selftest(self, n, reps):
        """`reps` back-to-back dot products over n doubles, in-kernel reduction tail against the two-launch form: mismatches"""
        bad = C.c_int(0)
        _chk(self.L.tp_grid_reduction_selftest(self.handle, int(n), int(reps), C.byref(bad)), "tp_grid_reduction_selftest")
        return bad.value

    def comm_timer(self, on):
        """N > 1: start / stop collecting the communication timing (tp_grid_comm_timer)"""
        _chk(self.L.tp_grid_comm_timer(self.handle, int(on)), "tp_grid_comm_timer")

    def comm_timer_read(self):
        """-> {kind: {"calls", "host_ms", "device_ms"}} for the kinds halo_blocking, halo_overlapped, all_reduce, all_gather"""
        n, h, d = (C.c_long * 4)(), (C.c_double * 4)(), (C.c_double * 4)()
        _chk(self.L.tp_grid_comm_timer_read(self.handle, n, h, d), "tp_grid_comm_timer_read")
        kinds = ("halo_blocking", "halo_overlapped", "all_reduce", "all_gather")
        return {k: {"calls": int(n[i]), "host_ms": float(h[i]), "device_ms": float(d[i])} for i, k in enumerate(kinds)}

    def kernel_timer(self, on):
        """bracket every launch of the fine level's fused Chebyshev step with a HIP event pair (bench.py's roofline)"""
        _chk(self.L.tp_grid_kernel_timer(self.handle, int(on)), "tp_grid_kernel_timer")

    def kernel_timer_read(self):
        """-> (average ms per launch, launches) since the timer was switched on / last read"""
        t, n = C.c_double(0.0), C.c_long(0)
        _chk(self.L.tp_grid_kernel_timer_read(self.handle, C.byref(t), C.byref(n)), "tp_grid_kernel_timer_read")
        return (t.value / n.value if n.value else 0.0), n.value

    def kernel_timer_read2(self):
        """-> (total ms, launches, algorithmic bytes of exactly those launches)"""
        t, n, b = C.c_double(0.0), C.c_long(0), C.c_double(0.0)
        _chk(self.L.tp_grid_kernel_timer_read2(self.handle, C.byref(t), C.byref(n), C.byref(b)), "tp_grid_kernel_timer_read2")
        return t.value, n.value, b.value

    def _use_rccl(self, group):
        """Hand the slab exchange to RCCL inside the library (tp_grid_use_rccl): same RCCL instance as
        torch.distributed's nccl backend, no Python round trip per halo.  Every rank takes the same decision."""
        import torch.distributed as dist
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        ok = os.path.exists(path) and self.L.tp_rccl_load(path.encode()) == 0
        idbuf, idbuf2 = C.create_string_buffer(128), C.create_string_buffer(128)
        if ok and self.part.rank == 0:   # two ids: collectives and neighbour exchanges on communicators of their own
            ok = self.L.tp_rccl_unique_id(idbuf) == 0 and self.L.tp_rccl_unique_id(idbuf2) == 0
        flag = torch.tensor([1 if ok else 0], device=self.device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag[0]) == 0:
            self._rccl_selfcheck = "not run: librccl.so could not be loaded on some rank -> torch.distributed hooks"
            return
        obj = [idbuf.raw + idbuf2.raw if self.part.rank == 0 else None]
        src = 0 if group is None else dist.get_global_rank(group, 0)
        dist.broadcast_object_list(obj, src=src, group=group)
        idb, idb2 = C.create_string_buffer(obj[0][:128], 128), C.create_string_buffer(obj[0][128:], 128)
        # one communicator by default: halo exchanges and all-reduces share it (stream order decides).  A second
        # communicator for the exchanges (TP_RCCL_TWO_COMM=1) lets them run beside the reductions, but two communicators
        # are only guaranteed to make progress concurrently if their kernels can co-reside -- opt-in until a run on two or
        # more GPUs has passed the self-check and a full solve with it
        if os.environ.get("TP_RCCL_TWO_COMM"):
            rc = self.L.tp_grid_use_rccl2(self.handle, idb, idb2)
        else:
            rc = self.L.tp_grid_use_rccl(self.handle, idb)
        flag = torch.tensor([1 if rc == 0 else 0], device=self.device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag[0]) == 0:   # some rank could not create its communicator: every rank goes back to the hooks
            self.L.tp_grid_drop_rccl(self.handle)
            print("topopt_amd: tp_grid_use_rccl failed on a rank; using the torch.distributed hooks", flush=True)
            self._rccl_selfcheck = "not run: ncclCommInitRank failed on some rank -> torch.distributed hooks"
            return
        # trust, but verify: rank-tagged planes through the new path; any rank unhappy -> everybody back to the hooks
        ok = C.c_int(0)
        rc = self.L.tp_grid_comm_selfcheck(self.handle, C.byref(ok))
        flag = torch.tensor([1 if (rc == 0 and ok.value == 1) else 0], device=self.device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag[0]) == 0:
            self.L.tp_grid_drop_rccl(self.handle)
            print("topopt_amd: in-library RCCL exchange failed its self-check; using the torch.distributed hooks", flush=True)
            self._rccl_selfcheck = "fail -> fallback to the torch.distributed hooks"
            return
        self._rccl_selfcheck = "pass"
        self.comm_kind = "rccl (in-library)"

    def _adopt(self, child):
        import weakref
        self._children.append(weakref.ref(child))

    def close(self):
        """Destroy the library objects NOW, dependents first (solvers, filters, optimisers built on this grid), after a
        device synchronisation -- the explicit, ordered counterpart of XxxDestroy in the reference (main.cc:126-134).
        Multi-rank programs call this on every rank before the process group goes (bench.py); idempotent."""
        for ref in list(getattr(self, "_children", ())):
            obj = ref()
            if obj is not None:
                obj.close()
        self._children = []
        if getattr(self, "handle", None):
            try:
                torch.cuda.synchronize(self.device)
            except Exception:
                pass
            self.L.tp_grid_destroy(self.handle)
            self.handle = None
        self.comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # sizes of the local arrays
    @property
    def n_local_nodes(self):
        return self.L.tp_grid_local_nodes(self.handle)

    @property
    def n_own_elems(self):
        return self.L.tp_grid_local_elems(self.handle)

    def node_vec(self, dof=3):
        return torch.zeros(self.n_local_nodes * dof, dtype=torch.float64, device=self.device)

    def elem_vec(self, value=0.0):
        return torch.full((self.n_own_elems,), value, dtype=torch.float64, device=self.device)

    def synth_density(self, seed=12345):
        x = self.elem_vec()
        _chk(self.L.tp_synth_density(self.handle, _ptr(x), seed), "tp_synth_density")
        return x

    def sync(self):
        _chk(self.L.tp_sync(self.handle), "tp_sync")


class LinearElasticity:
    """LinearElasticity (LinearElasticity.h:21-109) on the MI355X."""

    def __init__(self, grid, opts=None):
        self.grid, self.L = grid, grid.L
        self.opts = opts or SolverOptions()
        self.handle = C.c_void_p()
        self._o = self.opts.c_struct()
        _chk(self.L.tp_elasticity_create(C.byref(self.handle), grid.handle, C.byref(self._o)), "tp_elasticity_create")
        grid._adopt(self)
        self.U = grid.node_vec(3)       # state, persists across design iterations (warm start, :647)
        self.RHS = grid.node_vec(3)
        self.N = grid.node_vec(3)
        self.last_its, self.last_rnorm, self.last_bnorm, self.last_hist = 0, 0.0, 0.0, None

    def close(self):
        if getattr(self, "handle", None):
            self.L.tp_elasticity_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def KE(self):
        import numpy as np
        ke = np.zeros(576)
        _chk(self.L.tp_elasticity_get_ke(self.handle, ke.ctypes.data), "tp_elasticity_get_ke")
        return ke

    def KE_effective(self):
        """the element matrix the fine-level kernels apply inside the preconditioner (smoother, V-cycle residual), as a
        numpy.longdouble array of 576 (hi + lo of the library's double-double pair): KE in its packed block form (36 values),
        < 1 ulp of max|KE| away from KE"""
        import numpy as np
        hi, lo = np.zeros(576), np.zeros(576)
        _chk(self.L.tp_elasticity_get_ke_effective(self.handle, hi.ctypes.data, lo.ctypes.data), "tp_elasticity_get_ke_effective")
        return hi.astype(np.longdouble) + lo.astype(np.longdouble)

    def KE_krylov(self):
        """the element matrix of the Krylov operator (the plain products: A p and the initial residual of CG, MatMult): the packed
        form plus the translation mode's column and row of T KE T / 64 as KE has them; numpy.longdouble array of 576"""
        import numpy as np
        hi, lo = np.zeros(576), np.zeros(576)
        _chk(self.L.tp_elasticity_get_ke_krylov(self.handle, hi.ctypes.data, lo.ctypes.data), "tp_elasticity_get_ke_krylov")
        return hi.astype(np.longdouble) + lo.astype(np.longdouble)

    def SetUpLoadAndBC(self):
        """cantilever load case, LinearElasticity.cc:143-171"""
        _chk(self.L.tp_elasticity_cantilever(self.handle, _ptr(self.N), _ptr(self.RHS)), "tp_elasticity_cantilever")

    def SetUpLoadAndBC_MBB(self, load=-0.001):
        """Half MBB beam (BASELINE config 4; SURVEY D5: the reference ships only the cantilever, so this load case is
        data defined by the build): symmetry plane x = xmin (u_x = 0), roller along the edge x = xmax, z = zmin
        (u_z = 0, one node also u_y = 0), line load in -z along the edge x = xmin, z = zmax with half loads at the
        two end nodes.  Single rank or z-slabs (global z index decides)."""
        p = self.grid.part
        nx, ny, nzl = p.nx, p.ny, p.nz_local
        N = torch.ones(nzl, ny, nx, 3, dtype=torch.float64)
        R = torch.zeros(nzl, ny, nx, 3, dtype=torch.float64)
        kz = torch.arange(nzl) + p.node_z0
        N[:, :, 0, 0] = 0.0
        bot, top = kz == 0, kz == p.nz - 1
        N[bot, :, nx - 1, 2] = 0.0
        if bool(bot.any()):
            N[int(torch.nonzero(bot)[0]), 0, nx - 1, 1] = 0.0
        R[top, :, 0, 2] = load
        R[top, 0, 0, 2] = 0.5 * load
        R[top, ny - 1, 0, 2] = 0.5 * load
        self.SetBC(N.reshape(-1).to(self.U.device), R.reshape(-1).to(self.U.device))

    def SetBC(self, N, RHS):
        self.N.copy_(N)
        self.RHS.copy_(RHS)
        _chk(self.L.tp_elasticity_set_bc(self.handle, _ptr(self.N)), "tp_elasticity_set_bc")

    def AssembleStiffnessMatrix(self, xPhys, Emin, Emax, penal):
        _chk(self.L.tp_elasticity_assemble(self.handle, _ptr(xPhys), Emin, Emax, penal), "tp_elasticity_assemble")

    def MatMult(self, u, y=None):
        y = torch.zeros_like(u) if y is None else y
        _chk(self.L.tp_elasticity_apply(self.handle, _ptr(u), _ptr(y)), "tp_elasticity_apply")
        return y

    def MatMultKrylov(self, u, y=None):
        """the product CG multiplies with inside KSPSolve (the operator from KE_krylov(): KE's action to rounding on
        translation-dominated fields); MatMult applies the packed form (KE_effective())"""
        y = torch.zeros_like(u) if y is None else y
        _chk(self.L.tp_elasticity_apply_krylov(self.handle, _ptr(u), _ptr(y)), "tp_elasticity_apply_krylov")
        return y

    def KSPSolve(self, hist_cap=0):
        import numpy as np
        its, rn, bn = C.c_int(), C.c_double(), C.c_double()
        hist = np.zeros(max(hist_cap, 1))
        import time
        t0 = time.perf_counter()   # the solve ends with a host read of ||r||: wall time == device time
        rc = self.L.tp_elasticity_solve(self.handle, _ptr(self.RHS), _ptr(self.U), C.byref(its), C.byref(rn),
                                        C.byref(bn), hist.ctypes.data if hist_cap else None, hist_cap)
        self.last_solve_s = time.perf_counter() - t0
        self.last_its, self.last_rnorm, self.last_bnorm = its.value, rn.value, bn.value
        self.last_hist = hist[: min(its.value + 1, hist_cap)] if hist_cap else None
        _chk(rc, "tp_elasticity_solve")
        return its.value

    def SolveState(self, xPhys, Emin, Emax, penal, hist_cap=0):
        """LinearElasticity.cc:182-223"""
        self.AssembleStiffnessMatrix(xPhys, Emin, Emax, penal)
        return self.KSPSolve(hist_cap)

    def Objective(self, xPhys, Emin, Emax, penal, volfrac, dfdx=None, dgdx=None):
        fx, gx = C.c_double(), C.c_double()
        _chk(self.L.tp_elasticity_objective(self.handle, _ptr(self.U), _ptr(xPhys), Emin, Emax, penal, volfrac,
                                            C.byref(fx), C.byref(gx), _ptr(dfdx), _ptr(dgdx)),
             "tp_elasticity_objective")
        return fx.value, gx.value

    def ComputeObjectiveConstraintsSensitivities(self, dfdx, dgdx, xPhys, Emin, Emax, penal, volfrac, hist_cap=0):
        """LinearElasticity.cc:363-445 -> (fx, gx)"""
        self.SolveState(xPhys, Emin, Emax, penal, hist_cap)
        return self.Objective(xPhys, Emin, Emax, penal, volfrac, dfdx, dgdx)

    def ComputeObjectiveConstraints(self, xPhys, Emin, Emax, penal, volfrac, hist_cap=0):
        """LinearElasticity.cc:225-297: solve, then fx and gx -- no sensitivities -> (fx, gx)"""
        self.SolveState(xPhys, Emin, Emax, penal, hist_cap)
        fx, gx = C.c_double(), C.c_double()
        _chk(self.L.tp_elasticity_objective_only(self.handle, _ptr(self.U), _ptr(xPhys), Emin, Emax, penal, volfrac,
                                                 C.byref(fx), C.byref(gx)), "tp_elasticity_objective_only")
        return fx.value, gx.value

    def ComputeSensitivities(self, dfdx, dgdx, xPhys, Emin, Emax, penal, volfrac=0.0):
        """LinearElasticity.cc:299-361: dfdx, dgdx of the current state U (no solve)"""
        _chk(self.L.tp_elasticity_sensitivities(self.handle, _ptr(self.U), _ptr(xPhys), Emin, Emax, penal, _ptr(dfdx),
                                                _ptr(dgdx) if dgdx is not None else None), "tp_elasticity_sensitivities")

    # ---- introspection used by the parity tests ----------------------------
    def petsc_options(self):
        """The solver of the last assembly as a literal PETSc 3.11 option string (numeric Chebyshev windows per level)."""
        buf = C.create_string_buffer(8192)
        n = self.L.tp_elasticity_petsc_options(self.handle, buf, len(buf))
        if n < 0:
            raise TopOptError(2, "tp_elasticity_petsc_options (no assembly yet)")
        return buf.value.decode()

    def level_count(self):
        return self.L.tp_elasticity_level_count(self.handle)

    def level_nodes(self, l):
        return self.L.tp_elasticity_level_nodes(self.handle, l)

    def level_lambda(self, l):
        return self.L.tp_elasticity_level_lambda(self.handle, l)

    def level_lambda_min(self, l):
        return self.L.tp_elasticity_level_lambda_min(self.handle, l)

    def coarse_direct_active(self):
        """rows of the coarsest level if the last assembly factored it (SolverOptions.coarse_direct), else 0"""
        return self.L.tp_elasticity_coarse_direct_active(self.handle)

    def xcd_status(self):
        """process-wide: (recoveries from one-XCD kernels that gave up, one-XCD forms off, factorisation no longer deferred)"""
        import ctypes as C
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self.L.tp_xcd_status(C.byref(a), C.byref(b), C.byref(c))
        return a.value, bool(b.value), bool(c.value)

    def level_vec(self, l):
        return torch.zeros(3 * self.level_nodes(l), dtype=torch.float64, device=self.grid.device)

    def level_apply(self, l, u):
        y = torch.zeros_like(u)
        _chk(self.L.tp_elasticity_level_apply(self.handle, l, _ptr(u), _ptr(y)), "tp_elasticity_level_apply")
        return y

    def level_dinv(self, l):
        d = self.level_vec(l)
        _chk(self.L.tp_elasticity_level_diag(self.handle, l, _ptr(d)), "tp_elasticity_level_diag")
        return d

    def set_cycles(self, cycles):
        """PCMGSetCycleTypeOnLevel: cycles[l] cycles of level l + 1 per visit of level l (l = 0 finest; 1 = V, 2 = W)"""
        arr = (C.c_int * len(cycles))(*[int(v) for v in cycles])
        _chk(self.L.tp_elasticity_set_cycles(self.handle, arr, len(cycles)), "tp_elasticity_set_cycles")

    def level_pc(self, l, pc, r):
        """ksp_mode 1: z = M^-1 r on level l, pc 0 = PCJACOBI, 1 = PCSOR (one symmetric Gauss-Seidel sweep)"""
        z = torch.zeros_like(r)
        _chk(self.L.tp_elasticity_level_pc(self.handle, l, pc, _ptr(r), _ptr(z)), "tp_elasticity_level_pc")
        return z

    def level_gmres(self, l, pc, m, its, b, x, zero_guess=False, rtol=-1.0):
        """ksp_mode 1: the level's left-preconditioned GMRES(m), at most `its` iterations (rtol < 0: no test)"""
        import ctypes
        done = ctypes.c_int(0)
        _chk(self.L.tp_elasticity_level_gmres(self.handle, l, pc, m, its, rtol, _ptr(b), _ptr(x), int(zero_guess),
                                              ctypes.byref(done)), "tp_elasticity_level_gmres")
        return x, done.value

    def precond(self, r):
        z = torch.zeros_like(r)
        _chk(self.L.tp_elasticity_precond(self.handle, _ptr(r), _ptr(z)), "tp_elasticity_precond")
        return z

    def smooth(self, l, b, x, k, zero_guess=False):
        _chk(self.L.tp_elasticity_smooth(self.handle, l, _ptr(b), _ptr(x), k, int(zero_guess)), "tp_elasticity_smooth")
        return x

    def restrict(self, l, rf):
        rc = self.level_vec(l + 1)
        _chk(self.L.tp_elasticity_restrict(self.handle, l, _ptr(rf), _ptr(rc)), "tp_elasticity_restrict")
        return rc

    def prolong_add(self, l, xc, xf):
        _chk(self.L.tp_elasticity_prolong_add(self.handle, l, _ptr(xc), _ptr(xf)), "tp_elasticity_prolong_add")
        return xf

    def pop_stats(self):
        b, f, n = C.c_double(), C.c_double(), C.c_long()
        self.L.tp_elasticity_last_stats(self.handle, C.byref(b), C.byref(f), C.byref(n))
        return b.value, f.value, n.value


class Filter:
    """Filter (Filter.h:34-92): filterType 0 sensitivity, 1 density, 2 PDE."""

    def __init__(self, grid, filterType, rmin, pde_opts=None):
        self.grid, self.L = grid, grid.L
        self.filterType = filterType
        self.handle = C.c_void_p()
        po = pde_opts.c_struct() if pde_opts else None
        _chk(self.L.tp_filter_create(C.byref(self.handle), grid.handle, filterType, rmin,
                                     C.byref(po) if po else None), "tp_filter_create")
        grid._adopt(self)

    def close(self):
        if getattr(self, "handle", None):
            self.L.tp_filter_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ElemConn(self):
        return self.L.tp_filter_stencil_width(self.handle)

    def Hs(self):
        hs = self.grid.elem_vec()
        _chk(self.L.tp_filter_get_hs(self.handle, _ptr(hs)), "tp_filter_get_hs")
        return hs

    def KF(self):
        """PDE filter: the 8x8 Helmholtz element matrix (PDEFilter.cc:472-576), numpy"""
        import numpy as np
        kf = np.zeros(64)
        _chk(self.L.tp_filter_get_kf(self.handle, kf.ctypes.data_as(C.c_void_p)), "tp_filter_get_kf")
        return kf

    def FilterProject(self, x, xTilde, xPhys, projectionFilter=False, beta=0.1, eta=0.0):
        """Filter.cc:60-117"""
        _chk(self.L.tp_filter_project(self.handle, _ptr(x), _ptr(xTilde), _ptr(xPhys), int(projectionFilter), beta,
                                      eta), "tp_filter_project")

    def Gradients(self, x, xTilde, dfdx, dgdx, projectionFilter=False, beta=0.1, eta=0.0):
        """Filter.cc:120-204; dgdx is a list of tensors (m constraints)"""
        arr = (C.c_void_p * max(len(dgdx), 1))(*[_ptr(g) for g in dgdx])
        _chk(self.L.tp_filter_gradients(self.handle, _ptr(x), _ptr(xTilde), _ptr(dfdx), len(dgdx), arr,
                                        int(projectionFilter), beta, eta), "tp_filter_gradients")

    def MultH(self, x, y):
        """y = H x with the cone weights (MatMult(H, ...), Filter.cc:68, :173-189), no division by Hs"""
        _chk(self.L.tp_filter_mult_h(self.handle, _ptr(x), _ptr(y)), "tp_filter_mult_h")

    def PDEApply(self, u, y):
        """y = K_f u, the nodal Helmholtz operator of the PDE filter applied matrix-free (PDEFilter.cc:251-264 assembles it)"""
        _chk(self.L.tp_pdefilter_apply(self.handle, _ptr(u), _ptr(y)), "tp_pdefilter_apply")

    def GetMND(self, x):
        v = C.c_double()
        _chk(self.L.tp_filter_mnd(self.handle, _ptr(x), C.byref(v)), "tp_filter_mnd")
        return v.value

    def last_pde_solve(self):
        its, rn = C.c_int(), C.c_double()
        self.L.tp_filter_last_pde_its(self.handle, C.byref(its), C.byref(rn))
        return its.value, rn.value


class MMA:
    """MMA (MMA.h:29-140) on the device: the design vectors stay in HBM."""

    def __init__(self, grid, x, m=1, n_global=None):
        self.grid, self.L, self.m = grid, grid.L, m
        self.handle = C.c_void_p()
        n_loc = x.numel()
        n_glob = n_global if n_global is not None else grid.part.ex * grid.part.ey * grid.part.ez
        _chk(self.L.tp_mma_create(C.byref(self.handle), grid.handle, n_loc, n_glob, m, _ptr(x)), "tp_mma_create")
        grid._adopt(self)
        self.last_inner = 0

    def close(self):
        if getattr(self, "handle", None):
            self.L.tp_mma_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def SetOuterMovelimit(self, Xmin, Xmax, movlim, x, xmin, xmax):
        _chk(self.L.tp_mma_set_outer_movelimit(self.handle, Xmin, Xmax, movlim, _ptr(x), _ptr(xmin), _ptr(xmax)),
             "tp_mma_set_outer_movelimit")

    def Update(self, x, dfdx, gx, dgdx, xmin, xmax):
        g = (C.c_double * self.m)(*gx)
        dg = (C.c_void_p * self.m)(*[_ptr(t) for t in dgdx])
        it = C.c_int()
        _chk(self.L.tp_mma_update(self.handle, _ptr(x), _ptr(dfdx), g, dg, _ptr(xmin), _ptr(xmax), C.byref(it)),
             "tp_mma_update")
        self.last_inner = it.value

    def DesignChange(self, x, xold):
        ch = C.c_double()
        _chk(self.L.tp_mma_design_change(self.handle, _ptr(x), _ptr(xold), C.byref(ch)), "tp_mma_design_change")
        return ch.value

    def Restart(self, xo1, xo2, U, L):
        """MMA::Restart (MMA.cc:319-360): copy the two previous iterates and the asymptotes out"""
        _chk(self.L.tp_mma_restart_get(self.handle, _ptr(xo1), _ptr(xo2), _ptr(U), _ptr(L)), "tp_mma_restart_get")

    def SetRestart(self, k, xo1, xo2, U, L):
        """the restart constructor MMA::MMA(n, m, k, xo1, xo2, U, L, ...) (MMA.cc:22-106)"""
        _chk(self.L.tp_mma_restart_set(self.handle, int(k), _ptr(xo1), _ptr(xo2), _ptr(U), _ptr(L)), "tp_mma_restart_set")

    def state(self):
        lam = (C.c_double * self.m)()
        z, k = C.c_double(), C.c_int()
        self.L.tp_mma_get_state(self.handle, lam, C.byref(z), C.byref(k))
        return list(lam), z.value, k.value

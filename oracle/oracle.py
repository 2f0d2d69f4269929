"""ctypes binding of the CPU oracle (oracle/topopt_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# Two flavours of the SAME source: "f64" (the oracle: double, the reference's arithmetic type) and "ld" (the arbiter,
# oracle/arbiter.py: topopt_oracle.c with every `double` turned into x87 `long double` by the Makefile -- 64-bit
# mantissa, unit round-off 5.4e-20 -- used to decide which of two double-precision results is the closer one).  The
# arbiter module executes this file a second time with _FLAVOUR preset to "ld".
_FLAVOUR = globals().get("_FLAVOUR", "f64")
REAL = np.float64 if _FLAVOUR == "f64" else np.longdouble
c_real = C.c_double if _FLAVOUR == "f64" else C.c_longdouble
_SO = "liboracle.so" if _FLAVOUR == "f64" else "liboracle_ld.so"


def _z(n):
    return np.zeros(n, dtype=REAL)


def build(force=False):
    so = os.path.join(_HERE, _SO)
    srcs = [os.path.join(_HERE, f) for f in ("topopt_oracle.c", "mma_oracle.c", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, _SO], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        # many tiny OpenMP regions + more threads than usable cores (containers) is pathologically slow
        os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 16)))
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        _LIB = C.CDLL(build())
        L = _LIB
        L.orc_hash_u01.restype = c_real
        L.orc_hash_u01.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_elem_lambda_bound.restype = c_real
        L.orc_elem_lambda_bound.argtypes = [C.c_int, C.c_void_p]
        L.orc_mg_create.restype = C.c_void_p
        L.orc_mg_create.argtypes = [C.c_int] * 7 + [c_real] * 2
        L.orc_mg_destroy.argtypes = [C.c_void_p]
        L.orc_mg_set_fine_eig.argtypes = [C.c_void_p, C.c_int]
        L.orc_mg_set_nlanczos.argtypes = [C.c_void_p, C.c_int]
        L.orc_mg_set_coarse_direct.argtypes = [C.c_void_p, C.c_int]
        L.orc_mg_fine_matfree.argtypes = [C.c_void_p] * 4
        L.orc_mg_set_cycles.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_mg_assemble.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mg_reassemble_fine.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mg_set_krylov_operator.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mg_precond.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mg_solve.restype = C.c_int
        L.orc_mg_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, c_real, c_real, c_real, C.c_int,
                                   C.c_int, C.c_void_p, C.c_void_p]
        L.orc_mg_level_size.restype = C.c_long
        L.orc_mg_level_size.argtypes = [C.c_void_p, C.c_int]
        L.orc_mg_level_nnz.restype = C.c_long
        L.orc_mg_level_nnz.argtypes = [C.c_void_p, C.c_int]
        L.orc_mg_level_lambda.restype = c_real
        L.orc_mg_level_lambda.argtypes = [C.c_void_p, C.c_int]
        L.orc_mg_level_lambda_min.restype = c_real
        L.orc_mg_level_lambda_min.argtypes = [C.c_void_p, C.c_int]
        for f in ("orc_mg_level_apply", "orc_mg_prolong", "orc_mg_restrict"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_mg_level_diag.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_mg_level_csr.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mg_smooth.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_filter_create.restype = C.c_void_p
        L.orc_filter_create.argtypes = [C.c_int] * 3 + [c_real] * 4
        L.orc_filter_destroy.argtypes = [C.c_void_p]
        L.orc_filter_conn.argtypes = [C.c_void_p]
        L.orc_filter_nnz.restype = C.c_long
        L.orc_filter_nnz.argtypes = [C.c_void_p]
        L.orc_filter_hs.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_filter_project.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_real,
                                         c_real]
        L.orc_filter_gradient.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                          c_real, c_real]
        L.orc_mnd.restype = c_real
        L.orc_mnd.argtypes = [C.c_long, C.c_void_p]
        L.orc_heaviside.argtypes = [C.c_long, C.c_void_p, c_real, c_real, C.c_void_p]
        L.orc_heaviside_chain.argtypes = [C.c_long, C.c_void_p, c_real, c_real, C.c_void_p]
        L.orc_pdef_create.restype = C.c_void_p
        L.orc_pdef_create.argtypes = [C.c_int] * 3 + [c_real] * 4 + [C.c_int] * 3 + [c_real] * 2
        L.orc_pdef_destroy.argtypes = [C.c_void_p]
        L.orc_pdef_kf.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_pdef_apply.restype = C.c_int
        L.orc_pdef_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, c_real, C.c_int, C.c_void_p]
        L.orc_pdef_last_rnorm.restype = c_real
        L.orc_pdef_last_rnorm.argtypes = [C.c_void_p]
        L.orc_pdef_clamp.restype = C.c_long
        L.orc_pdef_clamp.argtypes = [C.c_long, C.c_void_p]
        if _FLAVOUR == "f64":   # (the arbiter library holds topopt_oracle.c only)
            L.orc_mma_create.restype = C.c_void_p
            L.orc_mma_create.argtypes = [C.c_long, C.c_int, C.c_void_p]
            L.orc_mma_destroy.argtypes = [C.c_void_p]
            L.orc_mma_set_device_order.argtypes = [C.c_void_p, C.c_int]
            L.orc_mma_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            L.orc_mma_outer_movelimit.argtypes = [C.c_long, c_real, c_real, c_real, C.c_void_p, C.c_void_p,
                                                  C.c_void_p]
            L.orc_mma_design_change.restype = c_real
            L.orc_mma_design_change.argtypes = [C.c_long, C.c_void_p, C.c_void_p]
            L.orc_mma_update.restype = C.c_int
            L.orc_mma_update.argtypes = [C.c_void_p] * 7
            L.orc_mma_kkt.argtypes = [C.c_void_p] * 9
        L.orc_simp.argtypes = [C.c_long, C.c_void_p, c_real, c_real, c_real, C.c_void_p]
        L.orc_synth_density.argtypes = [C.c_int] * 5 + [c_real, C.c_uint64, C.c_void_p]
        L.orc_matfree_apply.argtypes = [C.c_int] * 4 + [C.c_void_p] * 5
        L.orc_cantilever_bc.argtypes = [C.c_int] * 3 + [c_real] * 3 + [C.c_void_p] * 2
        L.orc_compliance_sens.argtypes = ([C.c_int] * 3 + [C.c_void_p] * 3 + [c_real] * 4 + [C.c_void_p] * 4)
        L.orc_hex8_ke.argtypes = [C.c_void_p] * 3 + [c_real, C.c_int, C.c_void_p]
        L.orc_hex8_ke_box.argtypes = [c_real] * 4 + [C.c_void_p]
        L.orc_pde_kf.argtypes = [c_real] * 4 + [C.c_void_p] * 2
    return _LIB


def _p(a):
    if a is None:
        return None
    assert a.dtype == REAL and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)   # (keeps a reference to `a`: a converted temporary lives until the call returns)


def f64(a):
    return np.ascontiguousarray(a, dtype=REAL)


def hex8_ke_box(dx, dy, dz, nu=0.3):
    ke = _z(576)
    lib().orc_hex8_ke_box(dx, dy, dz, nu, _p(ke))
    return ke


def pde_kf(dx, dy, dz, R):
    kf, tf = _z(64), _z(8)
    lib().orc_pde_kf(dx, dy, dz, R, _p(kf), _p(tf))
    return kf, tf


def elem_lambda_bound(ke):
    ed = int(round(np.sqrt(ke.size)))
    ke = f64(ke)
    return lib().orc_elem_lambda_bound(ed, _p(ke))


def cantilever_bc(nx, ny, nz, h):
    n = 3 * nx * ny * nz
    N, R = _z(n), _z(n)
    hx, hy, hz = (h, h, h) if np.isscalar(h) else h
    lib().orc_cantilever_bc(nx, ny, nz, hx, hy, hz, _p(N), _p(R))
    return N, R


def mbb_bc(nx, ny, nz, load=-0.001):
    """Half MBB beam (BASELINE config 4; SURVEY D5: the reference ships only the cantilever, this load case is data defined by
    the build -- the same data as api.LinearElasticity.SetUpLoadAndBC_MBB): symmetry plane x = xmin (u_x = 0), roller along the
    edge x = xmax, z = zmin (u_z = 0, one node also u_y = 0), line load in -z along the edge x = xmin, z = zmax with half loads
    at the two end nodes.  -> (N, RHS)"""
    N = np.ones((nz, ny, nx, 3))
    R = np.zeros((nz, ny, nx, 3))
    N[:, :, 0, 0] = 0.0
    N[0, :, nx - 1, 2] = 0.0
    N[0, 0, nx - 1, 1] = 0.0
    R[nz - 1, :, 0, 2] = load
    R[nz - 1, 0, 0, 2] = 0.5 * load
    R[nz - 1, ny - 1, 0, 2] = 0.5 * load
    return f64(N.reshape(-1)), f64(R.reshape(-1))


def simp(x, Emin=1e-9, Emax=1.0, penal=3.0):
    x = f64(x)
    E = np.zeros_like(x)
    lib().orc_simp(x.size, _p(x), Emin, Emax, penal, _p(E))
    return E


def synth_density(ex, ey, ez, h, seed=12345, e0z=0, ez_glob=None):
    x = _z(ex * ey * ez)
    lib().orc_synth_density(ex, ey, ez, e0z, ez_glob or ez, h, seed, _p(x))
    return x


def matfree_apply(nx, ny, nz, dof, KE, E, N, u):
    u = f64(u)
    y = np.zeros_like(u)
    lib().orc_matfree_apply(nx, ny, nz, dof, _p(f64(KE)), _p(E), _p(N), _p(u), _p(y))
    return y


def compliance_sens(nx, ny, nz, KE, U, xPhys, Emin=1e-9, Emax=1.0, penal=3.0, volfrac=0.12):
    nel = (nx - 1) * (ny - 1) * (nz - 1)
    fx, gx = c_real(), c_real()
    dfdx, dgdx = _z(nel), _z(nel)
    lib().orc_compliance_sens(nx, ny, nz, _p(f64(KE)), _p(f64(U)), _p(f64(xPhys)), Emin, Emax, penal, volfrac,
                              C.addressof(fx), C.addressof(gx), _p(dfdx), _p(dgdx))
    return fx.value, gx.value, dfdx, dgdx


class MG:
    """CG + Galerkin multigrid on assembled CSR matrices (the oracle solver)."""

    def __init__(self, nx, ny, nz, dof=3, nlv=3, nsmooth=4, ncoarse=30, cheb_lo=0.1, cheb_hi=1.1, fine_eig=0):
        self.L = lib()
        self.h = self.L.orc_mg_create(nx, ny, nz, dof, nlv, nsmooth, ncoarse, cheb_lo, cheb_hi)
        if not self.h:
            raise ValueError("mesh not coarsenable %d times" % (nlv - 1))
        self.L.orc_mg_set_fine_eig(self.h, fine_eig)
        self.nlv, self.dof = nlv, dof
        self.n = dof * nx * ny * nz

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_mg_destroy(self.h)
            self.h = None

    def set_nlanczos(self, n):
        """Lanczos steps of the smoothing levels' eigenvalue estimates (10 = PETSc's default); takes effect at the next assemble()"""
        self.L.orc_mg_set_nlanczos(self.h, int(n))

    def set_coarse_direct(self, on=True):
        """coarsest level solved exactly (banded Cholesky) instead of the Chebyshev run; takes effect at the next assemble()"""
        self.L.orc_mg_set_coarse_direct(self.h, int(on))

    def set_cycles(self, cycles):
        """cycles[l] cycles of level l + 1 per visit of level l (l = 0 finest): 1 = V, 2 = W (PCMGSetCycleTypeOnLevel)"""
        c = (C.c_int * 16)(*([int(v) for v in cycles] + [1] * 16)[:16])
        self.L.orc_mg_set_cycles(self.h, C.addressof(c))

    def assemble(self, KE, E=None, N=None):
        self.L.orc_mg_fine_matfree(None, None, None, None)
        self._keep = (f64(KE), None if E is None else f64(E), None if N is None else f64(N))
        self.L.orc_mg_assemble(self.h, *[_p(a) for a in self._keep])

    def reassemble_fine(self, KE):
        """diagnostic: only the fine-level operator (and its Jacobi diagonal) from another element matrix; the hierarchy of the
        last assemble() stays"""
        self._keep2 = (f64(KE), self._keep[1], self._keep[2])
        self.L.orc_mg_reassemble_fine(self.h, *[_p(a) for a in self._keep2])

    def set_krylov_operator(self, KE):
        """diagnostic: the Krylov method's own products (initial residual, A p) from another element matrix than the
        preconditioner's fine level (None: back to one operator); E and N of the last assemble()"""
        if KE is None:
            self.L.orc_mg_set_krylov_operator(self.h, None, None, None)
        else:
            self._keep3 = (f64(KE), self._keep[1], self._keep[2])
            self.L.orc_mg_set_krylov_operator(self.h, *[_p(a) for a in self._keep3])

    def fine_matfree(self, on=True):
        """CPU baseline variant: apply the fine-level operator of the solve matrix-free (OpenMP gather over the 8
        elements of a node) instead of the assembled CSR.  Call after assemble(); assemble() switches it off."""
        if on:
            self.L.orc_mg_fine_matfree(self.h, *[_p(a) for a in self._keep])
        else:
            self.L.orc_mg_fine_matfree(None, None, None, None)

    def size(self, l):
        return self.L.orc_mg_level_size(self.h, l)

    def lam(self, l):
        return self.L.orc_mg_level_lambda(self.h, l)

    def lam_min(self, l):
        return self.L.orc_mg_level_lambda_min(self.h, l)

    def apply(self, l, u):
        u = f64(u)
        y = np.zeros_like(u)
        self.L.orc_mg_level_apply(self.h, l, _p(u), _p(y))
        return y

    def diag(self, l):
        d = _z(self.size(l))
        self.L.orc_mg_level_diag(self.h, l, _p(d))
        return d

    def csr(self, l):
        import scipy.sparse as sp
        n, nnz = self.size(l), self.L.orc_mg_level_nnz(self.h, l)
        rp, ci, v = np.zeros(n + 1, dtype=np.int64), np.zeros(nnz, dtype=np.int32), _z(nnz)
        self.L.orc_mg_level_csr(self.h, l, rp.ctypes.data, ci.ctypes.data, _p(v))
        return sp.csr_matrix((v, ci, rp), shape=(n, n))

    def prolong(self, l, xc):
        xc = f64(xc)
        xf = _z(self.size(l))
        self.L.orc_mg_prolong(self.h, l, _p(xc), _p(xf))
        return xf

    def restrict(self, l, rf):
        rf = f64(rf)
        rc = _z(self.size(l + 1))
        self.L.orc_mg_restrict(self.h, l, _p(rf), _p(rc))
        return rc

    def smooth(self, l, b, x, k, zero_guess):
        b, x = f64(b), f64(x).copy()
        self.L.orc_mg_smooth(self.h, l, _p(b), _p(x), k, int(zero_guess))
        return x

    def precond(self, r):
        r = f64(r)
        z = np.zeros_like(r)
        self.L.orc_mg_precond(self.h, _p(r), _p(z))
        return z

    def solve(self, b, x0=None, rtol=1e-5, atol=1e-50, dtol=1e5, maxit=200, use_pc=True):
        b = f64(b)
        x = np.zeros_like(b) if x0 is None else f64(x0).copy()
        hist = _z(maxit + 1)
        rn = c_real()
        its = self.L.orc_mg_solve(self.h, _p(b), _p(x), rtol, atol, dtol, maxit, int(use_pc), _p(hist),
                                  C.addressof(rn))
        return x, its, hist[: max(its, 0) + 1].copy()


class Filter:
    def __init__(self, nx, ny, nz, h, rmin):
        self.L = lib()
        hx, hy, hz = (h, h, h) if np.isscalar(h) else h
        self.h = self.L.orc_filter_create(nx, ny, nz, hx, hy, hz, rmin)
        self.nel = (nx - 1) * (ny - 1) * (nz - 1)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_filter_destroy(self.h)
            self.h = None

    @property
    def conn(self):
        return self.L.orc_filter_conn(self.h)

    @property
    def nnz(self):
        return self.L.orc_filter_nnz(self.h)

    def hs(self):
        a = _z(self.nel)
        self.L.orc_filter_hs(self.h, _p(a))
        return a

    def project(self, ftype, x, proj=False, beta=0.1, eta=0.0):
        x = f64(x)
        xt, xp = np.zeros_like(x), np.zeros_like(x)
        self.L.orc_filter_project(self.h, ftype, _p(x), _p(xt), _p(xp), int(proj), beta, eta)
        return xt, xp

    def gradient(self, ftype, x, xTilde, df, proj=False, beta=0.1, eta=0.0):
        df = f64(df).copy()
        self.L.orc_filter_gradient(self.h, ftype, _p(f64(x)), _p(f64(xTilde)), _p(df), int(proj), beta, eta)
        return df


class PDEFilter:
    def __init__(self, nx, ny, nz, h, rmin, nlv=3, nsmooth=4, ncoarse=30, cheb_lo=0.1, cheb_hi=1.1):
        self.L = lib()
        hx, hy, hz = (h, h, h) if np.isscalar(h) else h
        self.h = self.L.orc_pdef_create(nx, ny, nz, hx, hy, hz, rmin, nlv, nsmooth, ncoarse, cheb_lo, cheb_hi)
        if not self.h:
            raise ValueError("mesh not coarsenable")
        self.nel = (nx - 1) * (ny - 1) * (nz - 1)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_pdef_destroy(self.h)
            self.h = None

    def apply(self, x, rtol=1e-8, maxit=60):
        x = f64(x)
        out = _z(self.nel)
        hist = _z(maxit + 1)
        its = self.L.orc_pdef_apply(self.h, _p(x), _p(out), rtol, maxit, _p(hist))
        return out, its, hist[: max(its, 0) + 1].copy()


def heaviside(xt, beta, eta):
    xt = f64(xt)
    y = np.zeros_like(xt)
    lib().orc_heaviside(xt.size, _p(xt), beta, eta, _p(y))
    return y


def heaviside_chain(xt, beta, eta):
    xt = f64(xt)
    y = np.zeros_like(xt)
    lib().orc_heaviside_chain(xt.size, _p(xt), beta, eta, _p(y))
    return y


def mnd(x):
    x = f64(x)
    return lib().orc_mnd(x.size, _p(x))


class MMA:
    """oracle MMA (mma_oracle.c); vectors are numpy arrays, dgdx a list of m arrays"""

    def __init__(self, x, m=1):
        self.L = lib()
        self.n, self.m = x.size, m
        self.h = self.L.orc_mma_create(self.n, m, _p(f64(x)))
        self.last_inner = 0

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_mma_destroy(self.h)
            self.h = None

    def set_device_order(self, on=True):
        """sum in the HIP kernels' order (workgroups of 256 over grid_for(n, 1024) groups, shuffle trees) and cube by
        multiplication: for bit-for-bit comparisons with the device; default = the reference's loops and pow()"""
        nb = min(max((self.n + 255) // 256, 1), 1024) if on else 0
        self.L.orc_mma_set_device_order(self.h, nb)

    def SetOuterMovelimit(self, Xmin, Xmax, movlim, x):
        xmin, xmax = _z(self.n), _z(self.n)
        self.L.orc_mma_outer_movelimit(self.n, Xmin, Xmax, movlim, _p(f64(x)), _p(xmin), _p(xmax))
        return xmin, xmax

    def Update(self, x, dfdx, gx, dgdx, xmin, xmax):
        """returns the new design"""
        xn = f64(x).copy()
        g = f64(np.asarray(gx, dtype=REAL))
        dg = f64(np.concatenate([f64(d) for d in dgdx]))
        self.last_inner = self.L.orc_mma_update(self.h, _p(xn), _p(f64(dfdx)), _p(g), _p(dg), _p(f64(xmin)),
                                                _p(f64(xmax)))
        return xn

    def DesignChange(self, x, xold):
        return self.L.orc_mma_design_change(self.n, _p(f64(x)), _p(xold))

    def state(self):
        lam = _z(self.m)
        z = c_real()
        self.L.orc_mma_get_state(self.h, _p(lam), C.addressof(z), None, None)
        return lam, z.value

    def kkt(self, x, dfdx, fx, dgdx, xmin, xmax):
        n2, ni = c_real(), c_real()
        dg = f64(np.concatenate([f64(d) for d in dgdx]))
        self.L.orc_mma_kkt(self.h, _p(f64(x)), _p(f64(dfdx)), _p(f64(np.asarray(fx, dtype=REAL))), _p(dg),
                           _p(f64(xmin)), _p(f64(xmax)), C.addressof(n2), C.addressof(ni))
        return n2.value, ni.value

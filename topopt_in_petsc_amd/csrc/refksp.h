// refksp.h -- the reference's own HARD-CODED solver configuration, run as written: a correctness mode on one device
// (tp_solver_opts::ksp_mode = 1; the fast path of this library is the CG / Chebyshev-Jacobi configuration of mg.h).
//
//   outer   KSPFGMRES(restart), right-preconditioned, KSPConvergedDefault on the recurrence residual with the reference
//           norm ||b|| of a non-zero initial guess                          (LinearElasticity.cc:620-650, PDEFilter.cc:276-285)
//   PC      PCMG, multiplicative V-cycle, Galerkin operators, Q1 transfer   (:697-707; PDEFilter.cc:327-335)
//   levels  KSPGMRES(nsmooth) for exactly nsmooth iterations (PCMG installs KSPConvergedSkip on its smoothers), left PC,
//           zero guess on the way down, the current iterate on the way up   (:734-746; PDEFilter.cc:366-378)
//   coarse  KSPGMRES(coarse_restart), at most ncoarse iterations, rtol coarse_rtol on the PRECONDITIONED residual
//           (left PC, zero guess)                                           (:720-731; PDEFilter.cc:350-363)
//   PC of the level solvers: PCSOR = MatSOR with PETSc's defaults (one LOCAL SYMMETRIC sweep, omega 1, zero guess: forward
//           then backward Gauss-Seidel in the natural row order) or PCJACOBI.
// GMRES orthogonalises with classical Gram-Schmidt, no refinement (PETSc's default for GMRES and FGMRES).
//
// The Gauss-Seidel sweeps are sequential in PETSc.  Here a sweep runs as (nx-1) + 2(ny-1) + 4(nz-1) + 1 launches: the
// nodes with i + 2j + 4k = t do not couple through a 27-point stencil and every lexicographically earlier neighbour of
// a node has a smaller t, so processing t = 0, 1, 2, ... in place reproduces the sequential sweep EXACTLY (same
// operands for every row), only ~1000x slower than the Chebyshev smoother -- which is why this is a correctness mode.
#pragma once
#include <vector>

#include "mg.h"

// one wavefront of an in-place Gauss-Seidel sweep; thread = (j, k) of an owned plane, i follows from t
template <int DOF, class Op>
__global__ __launch_bounds__(BLK) void k_gs_wave(Op op, const double *__restrict__ b, double *x, int t, int backward) {
    const Geom &g = op.g;
    const int np = g.own_hi - g.own_lo + 1;
    const long q = blockIdx.x * (long)BLK + threadIdx.x;
    if (q >= (long)g.ny * np) return;
    const int j = (int)(q % g.ny), kk = (int)(q / g.ny);
    const int i = t - 2 * j - 4 * kk;
    if (i < 0 || i >= g.nx) return;
    const int k = g.own_lo + kk;
    const long n = (long)i + (long)g.nx * (j + (long)g.ny * k);
    double y[DOF], D[DOF * DOF], xo[DOF], xn[DOF];
    op.apply(x, i, j, k, n, y);  // rows of this node with the current iterate (own old values included)
    op.diag_block(n, i, j, k, D);
#pragma unroll
    for (int r = 0; r < DOF; r++) xo[r] = xn[r] = x[n * DOF + r];
    // the node's own rows one after the other (ascending in a forward, descending in a backward sweep): a row sees the
    // dofs of this node that the sweep has already updated
#pragma unroll
    for (int s = 0; s < DOF; s++) {
        const int r = backward ? DOF - 1 - s : s;
        double ax = y[r];
#pragma unroll
        for (int c = 0; c < DOF; c++)
            if (backward ? c > r : c < r) ax = fma(D[r * DOF + c], xn[c] - xo[c], ax);
        xn[r] = xo[r] + (b[n * DOF + r] - ax) / D[r * DOF + r];
    }
#pragma unroll
    for (int r = 0; r < DOF; r++) x[n * DOF + r] = xn[r];
}

// out = a * in over [off, off + n)
__global__ __launch_bounds__(BLK) void k_scale_to(double *out, const double *in, double a, long off, long n) {
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) out[off + i] = a * in[off + i];
}

template <int DOF>
struct RefKsp {
    MGSolver<DOF> *mg = nullptr;
    tp_grid *grid = nullptr;
    static constexpr int CH = 8;  // basis vectors per allocation (FGMRES(100) grows as PETSc's does, not up front)
    static constexpr int NCOEF = 256, SLOT_NORM = 250;
    struct Basis {
        std::vector<double *> chunk;
        long nd = 0;
        double *vec(int j) const { return chunk[j / CH] + (long)(j % CH) * nd; }
    };
    Basis lvV[TP_MAX_LEVELS + 1];         // Krylov bases of the level solvers
    double *lvT[TP_MAX_LEVELS + 1] = {};  // level scratch: operator output / residual
    Basis oV, oZ;                         // outer FGMRES: Krylov basis and the preconditioned directions
    double *coef = nullptr, *part = nullptr, *hc = nullptr;  // device coefficients, reduction partials, pinned host copy
    long sweeps = 0;                      // Gauss-Seidel sweeps of the last solve (diagnostics)

    int ensure(Basis &B, int last, long nd) {
        B.nd = nd;
        while ((int)B.chunk.size() * CH <= last) {
            double *p = nullptr;
            TP_HIP(hipMalloc((void **)&p, sizeof(double) * (size_t)nd * CH));
            TP_HIP(hipMemsetAsync(p, 0, sizeof(double) * (size_t)nd * CH, grid->stream));
            B.chunk.push_back(p);
        }
        return TP_OK;
    }
    static void release(Basis &B) {
        for (double *p : B.chunk) (void)hipFree(p);
        B.chunk.clear();
    }
    int init() {
        TP_HIP(hipMalloc((void **)&coef, sizeof(double) * NCOEF));
        TP_HIP(hipMalloc((void **)&part, sizeof(double) * 256 * CH));
        TP_HIP(hipHostMalloc((void **)&hc, sizeof(double) * NCOEF));
        for (int l = 0; l < mg->nlv; l++) {
            const size_t nb = sizeof(double) * (size_t)mg->lv[l].ndof();
            TP_HIP(hipMalloc((void **)&lvT[l], nb));
            TP_HIP(hipMemsetAsync(lvT[l], 0, nb, grid->stream));
        }
        return TP_OK;
    }
    void free_all() {
        for (int l = 0; l <= TP_MAX_LEVELS; l++) {
            release(lvV[l]);
            (void)hipFree(lvT[l]);
            lvT[l] = nullptr;
        }
        release(oV);
        release(oZ);
        (void)hipFree(coef);
        (void)hipFree(part);
        (void)hipHostFree(hc);
        coef = part = hc = nullptr;
    }

    // ---- BLAS-1 on the owned range of level l -------------------------------------------------
    int read_coef(int first, int n, double *out) {
        TP_HIP(hipMemcpyAsync(hc + first, coef + first, sizeof(double) * n, hipMemcpyDeviceToHost, grid->stream));
        TP_HIP(hipStreamSynchronize(grid->stream));
        for (int i = 0; i < n; i++) out[i] = hc[first + i];
        return TP_OK;
    }
    // h[q] = V_q . w for q < nv: left in coef[0, nv) on the device and returned to the host
    int dots(int l, const Basis &B, int nv, const double *w, double *h) {
        Level<DOF> &L = mg->lv[l];
        const long off = L.own_off(), n = L.own_n();
        const int nb = n <= 65536 ? 1 : grid_for(n, 256);
        for (int c0 = 0; c0 < nv; c0 += CH) {
            const int cnt = nv - c0 < CH ? nv - c0 : CH;
            TP_LAUNCH(k_multi_dot, dim3(nb, cnt), dim3(BLK), 0, grid->stream, B.chunk[c0 / CH], B.nd, cnt, w, off, n,
                      nb == 1 ? coef + c0 : part, nullptr, nullptr);
            if (nb > 1) TP_LAUNCH(k_reduce_multi, dim3(cnt), dim3(BLK), 0, grid->stream, part, nb, cnt, coef + c0);
            grid->launches += nb > 1 ? 2 : 1;
        }
        TP_TRY(mg->allreduce_dev(coef, nv, L.no_comm));
        return read_coef(0, nv, h);
    }
    // w -= sum_q coef[q] V_q with the coefficients dots() left on the device
    int subtract(int l, const Basis &B, int nv, double *w) {
        Level<DOF> &L = mg->lv[l];
        const long off = L.own_off(), n = L.own_n();
        for (int c0 = 0; c0 < nv; c0 += CH) {
            const int cnt = nv - c0 < CH ? nv - c0 : CH;
            TP_LAUNCH(k_multi_axpy<false>, dim3(grid_for(n)), dim3(BLK), 0, grid->stream, B.chunk[c0 / CH], B.nd, cnt, coef + c0, w, off,
                      n, nullptr, nullptr, nullptr, nullptr, nullptr);
            count_launch(grid);
        }
        return TP_OK;
    }
    // x += sum_q y[q] B_q  (host coefficients)
    int combine(int l, const Basis &B, int nv, const double *y, double *x) {
        if (nv <= 0) return TP_OK;
        for (int q = 0; q < nv; q++) hc[q] = -y[q];
        TP_HIP(hipMemcpyAsync(coef, hc, sizeof(double) * nv, hipMemcpyHostToDevice, grid->stream));
        TP_TRY(subtract(l, B, nv, x));
        TP_HIP(hipStreamSynchronize(grid->stream));  // hc is reused by the next read
        return TP_OK;
    }
    int norm(int l, const double *w, double *out) {
        Level<DOF> &L = mg->lv[l];
        const long off = L.own_off(), n = L.own_n();
        const int nb = n <= 65536 ? 1 : grid_for(n, 256);
        TP_LAUNCH(k_multi_dot, dim3(nb, 1), dim3(BLK), 0, grid->stream, w, L.ndof(), 1, w, off, n, nb == 1 ? coef + SLOT_NORM : part, nullptr, nullptr);
        if (nb > 1) TP_LAUNCH(k_reduce_multi, dim3(1), dim3(BLK), 0, grid->stream, part, nb, 1, coef + SLOT_NORM);
        grid->launches += nb > 1 ? 2 : 1;
        TP_TRY(mg->allreduce_dev(coef + SLOT_NORM, 1, L.no_comm));
        double v;
        TP_TRY(read_coef(SLOT_NORM, 1, &v));
        *out = sqrt(v);
        return TP_OK;
    }
    int scale_to(int l, double *out, const double *in, double a) {
        Level<DOF> &L = mg->lv[l];
        TP_LAUNCH(k_scale_to, dim3(grid_for(L.own_n())), dim3(BLK), 0, grid->stream, out, in, a, L.own_off(), L.own_n());
        count_launch(grid);
        return TP_OK;
    }
    int resid(int l, const double *x, const double *b, double *r) {
        NodeArgs a{};
        a.x = x;
        a.out = r;
        a.b = b;
        TP_TRY(mg->halo(l, const_cast<double *>(x)));
        return mg->template op<EPI_RESID>(l, a);
    }

    // ---- z = M^-1 r : PCJACOBI (pc = 0) or PCSOR (pc = 1) ---------------------------------------
    template <class Op>
    int ssor(const Op &o, const double *r, double *z) {
        const Geom &g = o.g;
        const int np = g.own_hi - g.own_lo + 1;
        const int T = (g.nx - 1) + 2 * (g.ny - 1) + 4 * (np - 1) + 1;
        const int nb = (int)(((long)g.ny * np + BLK - 1) / BLK);
        for (int t = 0; t < T; t++) TP_LAUNCH((k_gs_wave<DOF, Op>), dim3(nb), dim3(BLK), 0, grid->stream, o, r, z, t, 0);
        for (int t = T - 1; t >= 0; t--) TP_LAUNCH((k_gs_wave<DOF, Op>), dim3(nb), dim3(BLK), 0, grid->stream, o, r, z, t, 1);
        grid->launches += 2 * T;
        sweeps += 2;
        return TP_OK;
    }
    int pc_apply(int l, int pc, const double *r, double *z) {
        Level<DOF> &L = mg->lv[l];
        if (pc == 0) {
            TP_LAUNCH(k_pw_mult, dim3(grid_for(L.own_n())), dim3(BLK), 0, grid->stream, z + L.own_off(), r + L.own_off(),
                      L.dinv + L.own_off(), L.own_n());
            count_launch(grid);
            return TP_OK;
        }
        // zero guess; the sweep is LOCAL (MatSOR_MPIAIJ: the diagonal block only) -- one device: the whole matrix
        TP_HIP(hipMemsetAsync(z, 0, sizeof(double) * (size_t)L.ndof(), grid->stream));
        if (L.kind == LV_MATFREE) return ssor(MatfreeOp<DOF>{L.KE, L.E, L.mask, L.g}, r, z);
        if (L.kind == LV_DIA) return ssor(DiaOp<DOF>{L.S, L.ndof(), L.g}, r, z);
        return TP_ERR_STATE;  // a level without rows (LV_MACRO is not built in this mode)
    }

    // ---- Hessenberg least squares on the host ---------------------------------------------------
    struct Hess {
        int m;
        std::vector<double> R, cs, sn, g;  // R: upper triangle after the rotations, column j at R[i * m + j]
        explicit Hess(int m_) : m(m_), R((size_t)(m_ + 1) * m_), cs(m_), sn(m_), g(m_ + 1) {}
        void start(double beta) {
            std::fill(g.begin(), g.end(), 0.0);
            g[0] = beta;
        }
        // column j = h[0..j+1] of the Arnoldi relation; returns the new residual norm |g[j+1]|
        double column(int j, double *h) {
            for (int i = 0; i < j; i++) {
                const double a = cs[i] * h[i] + sn[i] * h[i + 1];
                h[i + 1] = -sn[i] * h[i] + cs[i] * h[i + 1];
                h[i] = a;
            }
            const double d = hypot(h[j], h[j + 1]);
            cs[j] = d > 0.0 ? h[j] / d : 1.0;
            sn[j] = d > 0.0 ? h[j + 1] / d : 0.0;
            h[j] = d;
            g[j + 1] = -sn[j] * g[j];
            g[j] = cs[j] * g[j];
            for (int i = 0; i <= j; i++) R[(size_t)i * m + j] = h[i];
            return fabs(g[j + 1]);
        }
        void solve(int k, double *y) const {
            for (int i = k - 1; i >= 0; i--) {
                double s = g[i];
                for (int q = i + 1; q < k; q++) s -= R[(size_t)i * m + q] * y[q];
                y[i] = R[(size_t)i * m + i] != 0.0 ? s / R[(size_t)i * m + i] : 0.0;
            }
        }
    };

    // one Arnoldi step with classical Gram-Schmidt: w (= V_{j+1}, holding A-times-something) against V_0..V_j
    int arnoldi(int l, Basis &V, int j, double *h, double *hn) {
        double *w = V.vec(j + 1);
        TP_TRY(dots(l, V, j + 1, w, h));
        TP_TRY(subtract(l, V, j + 1, w));
        TP_TRY(norm(l, w, hn));
        h[j + 1] = *hn;
        if (*hn > 0.0) TP_TRY(scale_to(l, w, w, 1.0 / *hn));
        return TP_OK;
    }

    // ---- KSPGMRES with a left preconditioner on level l: x is updated in place -------------------
    int gmres(int l, const double *b, double *x, bool zero_guess, int m, int maxit, double rtol, double atol, double dtol,
              bool test, int pc, int *its_out) {
        Level<DOF> &L = mg->lv[l];
        if (m < 1) m = 1;
        Basis &V = lvV[l];
        TP_TRY(ensure(V, m, L.ndof()));
        double *t = lvT[l];
        if (zero_guess) TP_HIP(hipMemsetAsync(x, 0, sizeof(double) * (size_t)L.ndof(), grid->stream));
        Hess H(m);
        std::vector<double> h(m + 2), y(m);
        int its = 0;
        double rnorm0 = 0.0, ttol = 0.0;
        bool done = maxit < 1;
        while (!done) {
            if (zero_guess && its == 0) {
                TP_TRY(pc_apply(l, pc, b, V.vec(0)));
            } else {
                TP_TRY(resid(l, x, b, t));
                TP_TRY(pc_apply(l, pc, t, V.vec(0)));
            }
            double beta;
            TP_TRY(norm(l, V.vec(0), &beta));
            if (!(beta == beta)) return TP_ERR_DIVERGED;
            if (its == 0) {
                rnorm0 = beta;
                ttol = fmax(rtol * rnorm0, atol);
            }
            if (beta == 0.0 || (test && beta <= ttol) || its >= maxit) break;
            TP_TRY(scale_to(l, V.vec(0), V.vec(0), 1.0 / beta));
            H.start(beta);
            int j = 0;
            while (j < m && its < maxit) {
                TP_TRY(mg->apply(l, V.vec(j), t));
                TP_TRY(pc_apply(l, pc, t, V.vec(j + 1)));
                double hn;
                TP_TRY(arnoldi(l, V, j, h.data(), &hn));
                const double res = H.column(j, h.data());
                its++;
                j++;
                if (!(res == res)) return TP_ERR_DIVERGED;
                if (test && (res <= ttol || res >= dtol * rnorm0)) done = true;
                if (hn == 0.0) done = true;  // happy breakdown: the iterate is exact
                if (done) break;
            }
            if (its >= maxit) done = true;
            H.solve(j, y.data());
            TP_TRY(combine(l, V, j, y.data(), x));
        }
        if (its_out) *its_out = its;
        return TP_OK;
    }

    // ---- PCApply_MG: multiplicative V-cycle from a zero guess, result in lv[l].x ------------------
    int coarse_its = 0;
    int vcycle(int l, const double *b) {
        const tp_solver_opts &o = mg->opt;
        Level<DOF> &L = mg->lv[l];
        if (l == mg->nlv - 1) {
            int its = 0;
            TP_TRY(gmres(l, b, L.x, true, o.coarse_restart, o.ncoarse, o.coarse_rtol, o.atol, o.dtol, true, o.coarse_pc, &its));
            coarse_its += its;
            return TP_OK;
        }
        Level<DOF> &C = mg->lv[l + 1];
        TP_TRY(gmres(l, b, L.x, true, o.nsmooth, o.nsmooth, 0.0, 0.0, 0.0, false, o.smooth_pc, nullptr));
        TP_TRY(resid(l, L.x, b, L.r));
        TP_TRY(mg->halo(l, L.r));
        TP_LAUNCH((k_restrict<DOF>), dim3((int)((C.g.owned_nodes() + BLK - 1) / BLK)), dim3(BLK), 0, grid->stream, C.g, L.g, L.r,
                  C.b, nullptr, nullptr, nullptr, 0.0, 0L, -1L);
        count_launch(grid);
        TP_TRY(vcycle(l + 1, C.b));
        TP_TRY(mg->halo(l + 1, C.x));
        TP_LAUNCH((k_prolong_add<DOF>), dim3((int)((L.g.owned_nodes() + BLK - 1) / BLK)), dim3(BLK), 0, grid->stream, C.g, L.g,
                  C.x, L.x, 0L, -1L);
        count_launch(grid);
        return gmres(l, b, L.x, false, o.nsmooth, o.nsmooth, 0.0, 0.0, 0.0, false, o.smooth_pc, nullptr);
    }

    // ---- KSPSolve: FGMRES(restart) around the V-cycle --------------------------------------------
    int solve(const double *b, double *x, int *its_out, double *rnorm_out, double *bnorm_out, double *hist, int hist_cap) {
        const tp_solver_opts &o = mg->opt;
        Level<DOF> &L = mg->lv[0];
        const long nd = L.ndof();
        int m = o.restart < 1 ? 1 : o.restart;
        if (m > 200) m = 200;  // NCOEF
        sweeps = 0;
        coarse_its = 0;
        double bnorm;
        TP_TRY(norm(0, b, &bnorm));
        if (bnorm_out) *bnorm_out = bnorm;
        Hess H(m);
        std::vector<double> h(m + 2), y(m);
        int its = 0, rc = TP_OK;
        double res = 0.0, ref = 0.0, ttol = 0.0;
        bool done = false;
        TP_TRY(ensure(oV, 0, nd));
        while (!done) {
            TP_TRY(resid(0, x, b, oV.vec(0)));
            double beta;
            TP_TRY(norm(0, oV.vec(0), &beta));
            res = beta;
            if (!(beta == beta)) return TP_ERR_DIVERGED;
            if (its == 0) {
                // KSPConvergedDefault, non-zero initial guess: the reference norm is ||b|| (the residual's if b = 0)
                ref = bnorm > 0.0 ? bnorm : beta;
                ttol = fmax(o.rtol * ref, o.atol);
                if (hist && hist_cap > 0) hist[0] = beta;
            }
            if (beta <= ttol || its >= o.max_it) break;
            TP_TRY(scale_to(0, oV.vec(0), oV.vec(0), 1.0 / beta));
            H.start(beta);
            int j = 0;
            while (j < m && its < o.max_it) {
                TP_TRY(ensure(oV, j + 1, nd));
                TP_TRY(ensure(oZ, j, nd));
                TP_TRY(vcycle(0, oV.vec(j)));
                TP_HIP(hipMemcpyAsync(oZ.vec(j), L.x, sizeof(double) * (size_t)nd, hipMemcpyDeviceToDevice, grid->stream));
                TP_TRY(mg->apply(0, oZ.vec(j), oV.vec(j + 1)));
                double hn;
                TP_TRY(arnoldi(0, oV, j, h.data(), &hn));
                res = H.column(j, h.data());
                its++;
                j++;
                if (hist && its < hist_cap) hist[its] = res;
                if (!(res == res)) return TP_ERR_DIVERGED;
                if (res <= ttol) done = true;
                if (!(res <= o.dtol * ref)) {
                    rc = TP_ERR_DIVERGED;
                    done = true;
                }
                if (hn == 0.0) done = true;
                if (done) break;
            }
            if (its >= o.max_it) done = true;
            H.solve(j, y.data());
            TP_TRY(combine(0, oZ, j, y.data(), x));
        }
        TP_TRY(mg->drain_halos());
        if (its_out) *its_out = its;
        if (rnorm_out) *rnorm_out = res;
        return rc;
    }
};

template <int DOF>
int refksp_get(MGSolver<DOF> &mg, RefKsp<DOF> **out) {
    RefKsp<DOF> *R = static_cast<RefKsp<DOF> *>(mg.refksp);
    if (!R) {
        R = new RefKsp<DOF>();
        R->mg = &mg;
        R->grid = mg.grid;
        mg.refksp = R;
        const int rc = R->init();
        if (rc) return rc;
    }
    *out = R;
    return TP_OK;
}
// z = PCApply_MG(r): one V-cycle of this configuration; *z = lv[0].x
template <int DOF>
int refksp_precond(MGSolver<DOF> &mg, const double *r, double **z) {
    RefKsp<DOF> *R;
    TP_TRY(refksp_get(mg, &R));
    TP_TRY(R->vcycle(0, r));
    *z = mg.lv[0].x;
    return TP_OK;
}
template <int DOF>
int refksp_solve(MGSolver<DOF> &mg, const double *b, double *x, int *its, double *rnorm, double *bnorm, double *hist, int hist_cap) {
    if (mg.grid->has_comm) return TP_ERR_ARG;  // refused at creation already
    RefKsp<DOF> *R;
    TP_TRY(refksp_get(mg, &R));
    return R->solve(b, x, its, rnorm, bnorm, hist, hist_cap);
}
template <int DOF>
void refksp_free(MGSolver<DOF> &mg) {
    RefKsp<DOF> *R = static_cast<RefKsp<DOF> *>(mg.refksp);
    if (R) {
        R->free_all();
        delete R;
    }
    mg.refksp = nullptr;
}
template <int DOF>
void refksp_stats(const MGSolver<DOF> &mg, long *sweeps, int *coarse_its) {
    const RefKsp<DOF> *R = static_cast<const RefKsp<DOF> *>(mg.refksp);
    *sweeps = R ? R->sweeps : 0;
    *coarse_its = R ? R->coarse_its : 0;
}

// mma.h -- the optimizer step around the hot path on the device:
// Method of Moving Asymptotes with the dual interior-point sub-solver of the
// reference's MMA.cc (GenSub :522-649, SolveDIP :651-688, XYZofLAMBDA :690-740,
// DualGrad :742-777, DualHess :779-880, DualLineSearch :882-900, DualResidual :902-946).
// The design vectors never leave HBM: every pass is a streaming kernel with a
// deterministic block reduction; only the m x m Newton system (m = number of
// constraints, 1 in the reference) is solved on the host, as the reference does
// redundantly on every rank (:829-837, :948-981).
#pragma once

constexpr int MMA_MAXM = 8;
struct MmaLam {
    double lam[MMA_MAXM];
};

// GenSub: asymptotes, move limits, p/q coefficients; partials[j*nb + b] = sum pij/(U-x) + qij/(x-L)
__global__ __launch_bounds__(BLK) void k_mma_gensub(long n, int m, int k, double asyminit, double asymdec,
                                                    double asyminc, const double *__restrict__ x,
                                                    const double *__restrict__ xo1, const double *__restrict__ xo2,
                                                    const double *__restrict__ xmin, const double *__restrict__ xmax,
                                                    const double *__restrict__ dfdx, const double *const *dgdx,
                                                    double *__restrict__ L, double *__restrict__ U,
                                                    double *__restrict__ alpha, double *__restrict__ beta,
                                                    double *__restrict__ p0, double *__restrict__ q0,
                                                    double *__restrict__ pij, double *__restrict__ qij,
                                                    double *__restrict__ partials) {
#pragma clang fp contract(off)  // same unfused arithmetic as the reference's CPU loops
    double bs[MMA_MAXM];
    for (int j = 0; j < m; j++) bs[j] = 0.0;
    const double feps = 1.0e-6;
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) {
        const double xv = x[i], xmi_ = xmin[i], xma_ = xmax[i];
        double Li, Ui;
        if (k < 3) {
            Li = (xv + (-asyminit) * xma_) + asyminit * xmi_;
            Ui = (xv + asyminit * xma_) + (-asyminit) * xmi_;
        } else {
            const double x1 = xo1[i], x2 = xo2[i];
            const double helpvar = (xv - x1) * (x1 - x2);
            const double gamma = helpvar < 0.0 ? asymdec : (helpvar > 0.0 ? asyminc : 1.0);
            Li = xv - gamma * (x1 - L[i]);
            Ui = xv + gamma * (U[i] - x1);
            const double xmi = fmax(1.0e-5, xma_ - xmi_);
            Li = fmax(Li, xv - 10.0 * xmi);
            Li = fmin(Li, xv - 0.01 * xmi);
            Ui = fmax(Ui, xv + 0.01 * xmi);
            Ui = fmin(Ui, xv + 10.0 * xmi);
        }
        L[i] = Li;
        U[i] = Ui;
        alpha[i] = fmax(xmi_, 0.9 * Li + 0.1 * xv);
        beta[i] = fmin(xma_, 0.9 * Ui + 0.1 * xv);
        const double df = dfdx[i];
        const double ux = Ui - xv, xl = xv - Li;
        p0[i] = (ux * ux) * (fmax(0.0, df) + 0.001 * fabs(df) + 0.5 * feps / (Ui - Li));
        q0[i] = (xl * xl) * (fmax(0.0, -1.0 * df) + 0.001 * fabs(df) + 0.5 * feps / (Ui - Li));
        for (int j = 0; j < m; j++) {
            const double g = dgdx[j][i];
            const double pj = (ux * ux) * fmax(0.0, g), qj = (xl * xl) * fmax(0.0, -1.0 * g);
            pij[(long)j * n + i] = pj;
            qij[(long)j * n + i] = qj;
            bs[j] += pj / ux + qj / xl;
        }
    }
    for (int j = 0; j < m; j++) {
        const double t = block_sum(bs[j]);
        if (threadIdx.x == 0) partials[(long)j * gridDim.x + blockIdx.x] = t;
    }
}

// XYZofLAMBDA for x, fused with the sums DualGrad / DualResidual need and (HESS) the m x m dual Hessian.
// partials: [0,m) grad sums, [m, m + m*m) Hessian sums
template <int HESS>
__global__ __launch_bounds__(BLK) void k_mma_xyz(long n, int m, MmaLam lm, double *__restrict__ x,
                                                 const double *__restrict__ L, const double *__restrict__ U,
                                                 const double *__restrict__ alpha, const double *__restrict__ beta,
                                                 const double *__restrict__ p0, const double *__restrict__ q0,
                                                 const double *__restrict__ pij, const double *__restrict__ qij,
                                                 double *__restrict__ partials) {
#pragma clang fp contract(off)
    double gs[MMA_MAXM], hs[HESS ? MMA_MAXM * MMA_MAXM : 1];
    for (int j = 0; j < m; j++) gs[j] = 0.0;
    if (HESS)
        for (int j = 0; j < m * m; j++) hs[j] = 0.0;
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) {
        const double Li = L[i], Ui = U[i];
        double pj = p0[i], qj = q0[i];
        double pv[MMA_MAXM], qv[MMA_MAXM];
        for (int j = 0; j < m; j++) {
            pv[j] = pij[(long)j * n + i];
            qv[j] = qij[(long)j * n + i];
            pj += pv[j] * lm.lam[j];
            qj += qv[j] * lm.lam[j];
        }
        const double sp = sqrt(pj), sq = sqrt(qj);
        const double xp = (sp * Li + sq * Ui) / (sp + sq);
        double xv = xp;
        if (xv < alpha[i]) xv = alpha[i];
        if (xv > beta[i]) xv = beta[i];
        x[i] = xv;
        const double ux = Ui - xv, xl = xv - Li;
        for (int j = 0; j < m; j++) gs[j] += pv[j] / ux + qv[j] / xl;
        if (HESS) {
            double df2 = -1.0 / (2.0 * pj / (ux * ux * ux) + 2.0 * qj / (xl * xl * xl));
            if (xp < alpha[i]) df2 = 0.0;
            if (xp > beta[i]) df2 = 0.0;
            double PQ[MMA_MAXM];
            for (int j = 0; j < m; j++) PQ[j] = pv[j] / (ux * ux) - qv[j] / (xl * xl);
            for (int a = 0; a < m; a++)
                for (int b = 0; b < m; b++) hs[a * m + b] += (PQ[a] * df2) * PQ[b];
        }
    }
    for (int j = 0; j < m; j++) {
        const double t = block_sum(gs[j]);
        if (threadIdx.x == 0) partials[(long)j * gridDim.x + blockIdx.x] = t;
    }
    if (HESS)
        for (int j = 0; j < m * m; j++) {
            const double t = block_sum(hs[j]);
            if (threadIdx.x == 0) partials[(long)(m + j) * gridDim.x + blockIdx.x] = t;
        }
}

__global__ __launch_bounds__(BLK) void k_mma_movelimit(long n, double Xmin, double Xmax, double movlim,
                                                       const double *__restrict__ x, double *__restrict__ xmin,
                                                       double *__restrict__ xmax) {
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) {
        xmax[i] = fmin(Xmax, x[i] + movlim);
        xmin[i] = fmax(Xmin, x[i] - movlim);
    }
}
// partial max |x - xold|, then xold <- x
__global__ __launch_bounds__(BLK) void k_mma_change(long n, const double *__restrict__ x, double *__restrict__ xold,
                                                    double *__restrict__ partials) {
    __shared__ double s_m[BLK];
    double ch = 0.0;
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) {
        ch = fmax(ch, fabs(x[i] - xold[i]));
        xold[i] = x[i];
    }
    s_m[threadIdx.x] = ch;
    __syncthreads();
    for (int o = BLK / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) s_m[threadIdx.x] = fmax(s_m[threadIdx.x], s_m[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = s_m[0];
}
__global__ __launch_bounds__(BLK) void k_max_final(const double *__restrict__ partials, int nb, double *__restrict__ out) {
    __shared__ double s_m[BLK];
    double v = 0.0;
    for (int b = threadIdx.x; b < nb; b += BLK) v = fmax(v, partials[b]);
    s_m[threadIdx.x] = v;
    __syncthreads();
    for (int o = BLK / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) s_m[threadIdx.x] = fmax(s_m[threadIdx.x], s_m[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s_m[0];
}

struct tp_mma {
    tp_grid *grid;
    long n, nglob;
    int m, k;
    double asyminit, asymdec, asyminc;
    double a[MMA_MAXM], c[MMA_MAXM], y[MMA_MAXM], lam[MMA_MAXM], mu[MMA_MAXM], b[MMA_MAXM];
    double z;
    double *L, *U, *alpha, *beta, *p0, *q0, *pij, *qij, *xo1, *xo2;
    const double **d_dgdx;  // [dev] m pointers
    double *red;            // [dev] m + m*m reduced values
    double *part;           // [dev] (m + m*m) x 1024 block partials of the MMA kernels (own buffer: g->partials is
                            //       sized for 4 values per block)
    int last_inner;
};

static int mma_reduce(tp_mma *M, int nb, int nv, double *host) {
    tp_grid *g = M->grid;
    TP_LAUNCH(k_reduce_multi, dim3(nv), dim3(BLK), 0, g->stream, M->part, nb, nv, M->red);
    count_launch(g);
    if (g->has_comm)
        for (int o = 0; o < nv; o += 16) {
            const int cnt = nv - o < 16 ? nv - o : 16;
            TP_HIP(hipMemcpyAsync(g->comm.red, M->red + o, sizeof(double) * cnt, hipMemcpyDeviceToDevice, g->stream));
            if (g->comm.allreduce_sum(g->comm.user, cnt)) return TP_ERR_COMM;
            TP_HIP(hipMemcpyAsync(M->red + o, g->comm.red, sizeof(double) * cnt, hipMemcpyDeviceToDevice, g->stream));
        }
    TP_HIP(hipMemcpyAsync(g->h_scal, M->red, sizeof(double) * nv, hipMemcpyDeviceToHost, g->stream));
    TP_HIP(hipStreamSynchronize(g->stream));
    for (int i = 0; i < nv; i++) host[i] = g->h_scal[i];
    return TP_OK;
}

extern "C" int tp_mma_create(tp_mma **out, tp_grid *g, long n_local, long n_global, int m, const double *x) {
    if (!out || !g || m < 1 || m > MMA_MAXM || m + m * m > 64) return TP_ERR_ARG;
    tp_mma *M = new tp_mma();
    M->grid = g;
    M->n = n_local;
    M->nglob = n_global;
    M->m = m;
    M->k = 0;
    M->asyminit = 0.5;  // MMA.cc:31-33
    M->asymdec = 0.7;
    M->asyminc = 1.2;
    M->z = 0.0;
    for (int j = 0; j < m; j++) {
        M->a[j] = 0.0;  // MMA.cc:129-130
        M->c[j] = 1000.0;
        M->y[j] = M->lam[j] = M->mu[j] = M->b[j] = 0.0;
    }
    const size_t nb = sizeof(double) * (size_t)n_local;
    for (double **p : {&M->L, &M->U, &M->alpha, &M->beta, &M->p0, &M->q0, &M->xo1, &M->xo2}) {
        TP_HIP(hipMalloc((void **)p, nb));
        TP_HIP(hipMemsetAsync(*p, 0, nb, g->stream));
    }
    TP_HIP(hipMalloc((void **)&M->pij, nb * m));
    TP_HIP(hipMalloc((void **)&M->qij, nb * m));
    TP_HIP(hipMalloc((void **)&M->d_dgdx, sizeof(double *) * m));
    TP_HIP(hipMalloc((void **)&M->red, sizeof(double) * 128));
    TP_HIP(hipMalloc((void **)&M->part, sizeof(double) * 1024 * (size_t)(m + m * m)));
    TP_HIP(hipMemcpyAsync(M->xo1, x, nb, hipMemcpyDeviceToDevice, g->stream));
    TP_HIP(hipMemcpyAsync(M->xo2, x, nb, hipMemcpyDeviceToDevice, g->stream));
    M->last_inner = 0;
    *out = M;
    return TP_OK;
}
extern "C" int tp_mma_destroy(tp_mma *M) {
    if (!M) return TP_OK;
    (void)hipStreamSynchronize(M->grid->stream);
    for (void *p : {(void *)M->L, (void *)M->U, (void *)M->alpha, (void *)M->beta, (void *)M->p0, (void *)M->q0,
                    (void *)M->xo1, (void *)M->xo2, (void *)M->pij, (void *)M->qij, (void *)M->d_dgdx, (void *)M->red, (void *)M->part})
        (void)hipFree(p);
    delete M;
    return TP_OK;
}
extern "C" int tp_mma_set_outer_movelimit(tp_mma *M, double Xmin, double Xmax, double movlim, const double *x,
                                          double *xmin, double *xmax) {
    TP_LAUNCH(k_mma_movelimit, dim3(grid_for(M->n)), dim3(BLK), 0, M->grid->stream, M->n, Xmin, Xmax, movlim, x,
                       xmin, xmax);
    count_launch(M->grid, 24.0 * M->n, 2.0 * M->n);
    return TP_OK;
}
extern "C" int tp_mma_design_change(tp_mma *M, const double *x, double *xold, double *ch) {
    tp_grid *g = M->grid;
    const int nb = grid_for(M->n, 1024);
    TP_LAUNCH(k_mma_change, dim3(nb), dim3(BLK), 0, g->stream, M->n, x, xold, M->part);
    TP_LAUNCH(k_max_final, dim3(1), dim3(BLK), 0, g->stream, M->part, nb, M->red);
    count_launch(g, 24.0 * M->n, 1.0 * M->n);
    TP_HIP(hipMemcpyAsync(g->h_scal, M->red, sizeof(double), hipMemcpyDeviceToHost, g->stream));
    TP_HIP(hipStreamSynchronize(g->stream));
    double v = g->h_scal[0];
    if (g->has_comm) {
        // max over ranks through the sum hook: every rank's value in its own slot, 16 slots (the hook's buffer) at a time
        const double mine = v;
        for (int o = 0; o < g->nranks; o += 16) {
            const int cnt = g->nranks - o < 16 ? g->nranks - o : 16;
            double slots[16] = {0};
            if (g->rank >= o && g->rank < o + cnt) slots[g->rank - o] = mine;
            TP_HIP(hipMemcpyAsync(g->comm.red, slots, sizeof(double) * cnt, hipMemcpyHostToDevice, g->stream));
            TP_HIP(hipStreamSynchronize(g->stream));  // `slots` is a stack buffer
            if (g->comm.allreduce_sum(g->comm.user, cnt)) return TP_ERR_COMM;
            TP_HIP(hipMemcpyAsync(g->h_scal, g->comm.red, sizeof(double) * cnt, hipMemcpyDeviceToHost, g->stream));
            TP_HIP(hipStreamSynchronize(g->stream));
            for (int r = 0; r < cnt; r++) v = fmax(v, g->h_scal[r]);
        }
    }
    *ch = v;
    return TP_OK;
}

static void mma_lu_factorize(double *K, int nn) {
    for (int ss = 0; ss < nn - 1; ss++)
        for (int i = ss + 1; i < nn; i++) {
            K[i * nn + ss] = K[i * nn + ss] / K[ss * nn + ss];
            for (int j = ss + 1; j < nn; j++) K[i * nn + j] = K[i * nn + j] - K[i * nn + ss] * K[ss * nn + j];
        }
}
static void mma_lu_solve(const double *K, double *x, int nn) {
    for (int i = 1; i < nn; i++) {
        double a = 0.0;
        for (int j = 0; j < i; j++) a = a - K[i * nn + j] * x[j];
        x[i] = x[i] + a;
    }
    x[nn - 1] = x[nn - 1] / K[(nn - 1) * nn + (nn - 1)];
    for (int i = nn - 2; i >= 0; i--) {
        double a = x[i];
        for (int j = i + 1; j < nn; j++) a = a - K[i * nn + j] * x[j];
        x[i] = a / K[i * nn + i];
    }
}

// MMA::Update (MMA.cc:499-518).  x, dfdx, xmin, xmax, dgdx[j] [dev, n]; gx host (m).
extern "C" int tp_mma_update(tp_mma *M, double *x, const double *dfdx, const double *gx, const double *const *dgdx,
                             const double *xmin, const double *xmax, int *inner_its) {
    tp_grid *g = M->grid;
    hipStream_t s = g->stream;
    const long n = M->n;
    const int m = M->m;
    const int nb = grid_for(n, 1024);
    double red[64];
    TP_HIP(hipMemcpyAsync(M->d_dgdx, dgdx, sizeof(double *) * m, hipMemcpyHostToDevice, s));
    // ---- GenSub
    M->k++;
    TP_LAUNCH(k_mma_gensub, dim3(nb), dim3(BLK), 0, s, n, m, M->k, M->asyminit, M->asymdec, M->asyminc, x, M->xo1,
                       M->xo2, xmin, xmax, dfdx, M->d_dgdx, M->L, M->U, M->alpha, M->beta, M->p0, M->q0, M->pij, M->qij,
                       M->part);
    count_launch(g, 8.0 * n * (12.0 + 3.0 * m), 40.0 * n);
    TP_TRY(mma_reduce(M, nb, m, red));
    for (int j = 0; j < m; j++) M->b[j] = red[j] - gx[j];
    TP_HIP(hipMemcpyAsync(M->xo2, M->xo1, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
    TP_HIP(hipMemcpyAsync(M->xo1, x, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
    // ---- SolveDIP
    for (int j = 0; j < m; j++) {
        M->lam[j] = M->c[j] / 2.0;
        M->mu[j] = 1.0;
    }
    const double tol = 1.0e-9 * sqrt((double)(m + M->nglob));
    double epsi = 1.0, err = 1.0;
    double grad[MMA_MAXM], sv[2 * MMA_MAXM], Hess[MMA_MAXM * MMA_MAXM];
    int total = 0;
    auto lam_yz = [&](MmaLam &lm) {  // the scalar part of XYZofLAMBDA (:706-713)
        double lamai = 0.0;
        for (int j = 0; j < m; j++) {
            if (M->lam[j] < 0.0) M->lam[j] = 0;
            M->y[j] = fmax(0.0, M->lam[j] - M->c[j]);
            lamai += M->lam[j] * M->a[j];
            lm.lam[j] = M->lam[j];
        }
        M->z = fmax(0.0, 10.0 * (lamai - 1.0));
    };
    while (epsi > tol) {
        int loop = 0;
        while (err > 0.9 * epsi && loop < 100) {
            loop++;
            total++;
            MmaLam lm;
            lam_yz(lm);
            TP_LAUNCH((k_mma_xyz<1>), dim3(nb), dim3(BLK), 0, s, n, m, lm, x, M->L, M->U, M->alpha, M->beta, M->p0,
                               M->q0, M->pij, M->qij, M->part);
            count_launch(g, 8.0 * n * (7.0 + 2.0 * m), 40.0 * n);
            TP_TRY(mma_reduce(M, nb, m + m * m, red));
            for (int j = 0; j < m; j++) {
                grad[j] = red[j] - M->b[j] - M->a[j] * M->z - M->y[j];  // DualGrad
                grad[j] = -1.0 * grad[j] - epsi / M->lam[j];
            }
            // DualHess tail (:840-866)
            for (int j = 0; j < m * m; j++) Hess[j] = red[m + j];
            double lamai = 0.0;
            for (int j = 0; j < m; j++) {
                if (M->lam[j] < 0.0) M->lam[j] = 0.0;
                lamai += M->lam[j] * M->a[j];
                if (M->lam[j] > M->c[j]) Hess[j * m + j] += -1.0;
                Hess[j * m + j] += -M->mu[j] / M->lam[j];
            }
            if (lamai > 0.0)
                for (int j = 0; j < m; j++)
                    for (int kk = 0; kk < m; kk++) Hess[j * m + kk] += -10.0 * M->a[j] * M->a[kk];
            double tr = 0.0;
            for (int i = 0; i < m; i++) tr += Hess[i * m + i];
            double corr = 1e-4 * tr / m;
            if (-1.0 * corr < 1.0e-7) corr = -1.0e-7;
            for (int i = 0; i < m; i++) Hess[i * m + i] += corr;
            mma_lu_factorize(Hess, m);
            mma_lu_solve(Hess, grad, m);
            for (int j = 0; j < m; j++) sv[j] = grad[j];
            for (int i = 0; i < m; i++) sv[m + i] = -M->mu[i] + epsi / M->lam[i] - sv[i] * M->mu[i] / M->lam[i];
            double theta = 1.005;  // DualLineSearch
            for (int i = 0; i < m; i++) {
                if (theta < -1.01 * sv[i] / M->lam[i]) theta = -1.01 * sv[i] / M->lam[i];
                if (theta < -1.01 * sv[i + m] / M->mu[i]) theta = -1.01 * sv[i + m] / M->mu[i];
            }
            theta = 1.0 / theta;
            for (int i = 0; i < m; i++) {
                M->lam[i] = M->lam[i] + theta * sv[i];
                M->mu[i] = M->mu[i] + theta * sv[i + m];
            }
            lam_yz(lm);
            TP_LAUNCH((k_mma_xyz<0>), dim3(nb), dim3(BLK), 0, s, n, m, lm, x, M->L, M->U, M->alpha, M->beta, M->p0,
                               M->q0, M->pij, M->qij, M->part);
            count_launch(g, 8.0 * n * (7.0 + 2.0 * m), 20.0 * n);
            TP_TRY(mma_reduce(M, nb, m, red));
            err = 0.0;  // DualResidual
            for (int j = 0; j < m; j++) {
                const double r1 = red[j] - M->b[j] - M->a[j] * M->z - M->y[j] + M->mu[j];
                const double r2 = M->mu[j] * M->lam[j] - epsi;
                err = fmax(err, fmax(fabs(r1), fabs(r2)));
            }
        }
        epsi = epsi * 0.1;
    }
    M->last_inner = total;
    if (inner_its) *inner_its = total;
    return TP_OK;
}
extern "C" int tp_mma_get_state(const tp_mma *M, double *lam, double *z, int *k) {
    if (lam)
        for (int j = 0; j < M->m; j++) lam[j] = M->lam[j];
    if (z) *z = M->z;
    if (k) *k = M->k;
    return TP_OK;
}

extern "C" int tp_mma_restart_get(const tp_mma *M, double *xo1, double *xo2, double *U, double *L) {
    if (!M || !xo1 || !xo2 || !U || !L) return TP_ERR_ARG;
    hipStream_t st = M->grid->stream;
    const size_t nb = sizeof(double) * M->n;
    TP_HIP(hipMemcpyAsync(xo1, M->xo1, nb, hipMemcpyDeviceToDevice, st));
    TP_HIP(hipMemcpyAsync(xo2, M->xo2, nb, hipMemcpyDeviceToDevice, st));
    TP_HIP(hipMemcpyAsync(U, M->U, nb, hipMemcpyDeviceToDevice, st));
    TP_HIP(hipMemcpyAsync(L, M->L, nb, hipMemcpyDeviceToDevice, st));
    return TP_OK;
}
extern "C" int tp_mma_restart_set(tp_mma *M, int k, const double *xo1, const double *xo2, const double *U,
                                  const double *L) {
    if (!M || k < 0 || !xo1 || !xo2 || !U || !L) return TP_ERR_ARG;
    hipStream_t st = M->grid->stream;
    const size_t nb = sizeof(double) * M->n;
    TP_HIP(hipMemcpyAsync(M->xo1, xo1, nb, hipMemcpyDeviceToDevice, st));
    TP_HIP(hipMemcpyAsync(M->xo2, xo2, nb, hipMemcpyDeviceToDevice, st));
    TP_HIP(hipMemcpyAsync(M->U, U, nb, hipMemcpyDeviceToDevice, st));
    TP_HIP(hipMemcpyAsync(M->L, L, nb, hipMemcpyDeviceToDevice, st));
    M->k = k;
    return TP_OK;
}

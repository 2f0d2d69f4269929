/*
 * mma_oracle.c -- CPU ORACLE (test infrastructure) for the optimizer step that
 * calls the hot path: the Method of Moving Asymptotes with the dual
 * interior-point sub-solver of Aage & Lazarov (2013), restated from the
 * behaviour of the reference's MMA.cc (serial, one partition):
 *   GenSub :522-649, SolveDIP :651-688, XYZofLAMBDA :690-740, DualGrad :742-777,
 *   DualHess :779-880, DualLineSearch :882-900, DualResidual :902-946,
 *   Factorize/Solve :948-981, SetOuterMovelimit :386-405, DesignChange :407-426,
 *   KKTresidual :428-496.
 * The reference ships no test for MMA and needs PETSc Vecs to run: parity of this
 * restatement is anchored on optimality/feasibility properties (tests/test_mma.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

typedef struct {
    long n;
    int m, k;
    double asyminit, asymdec, asyminc;
    double *a, *c, *d, *y, *lam, *mu, *b, *grad, *s, *Hess;
    double z;
    double *L, *U, *alpha, *beta, *p0, *q0, *pij, *qij, *xo1, *xo2; /* pij/qij: m blocks of n */
    int dev_order; /* 0: the reference's left-to-right sums and pow(); > 0: the DEVICE's operation order (workgroups) */
} orc_mma_t;

static double dmin(double a, double b) { return a < b ? a : b; }
static double dmax(double a, double b) { return a > b ? a : b; }
static double dabs(double a) { return a > 0 ? a : -1.0 * a; }

/* ---- optional: the summation order of the HIP implementation (csrc/common.h: block_sum, csrc/mg.h:
 * k_reduce_multi), so that device and oracle can be compared BIT FOR BIT.  nb workgroups of 256 threads stride
 * through the terms; per workgroup a 64-lane shuffle-down tree per wave, then the 4 wave sums left to right;
 * the workgroup partials are summed by one workgroup in the same way.  The default (dev_order = 0) is the
 * reference's plain left-to-right loop. */
static double dev_block_sum(const double *v256) {
    double w[4];
    for (int wv = 0; wv < 4; wv++) {
        double l[64];
        for (int i = 0; i < 64; i++) l[i] = v256[wv * 64 + i];
        for (int off = 32; off > 0; off >>= 1)
            for (int i = 0; i + off < 64; i++) l[i] = l[i] + l[i + off];
        w[wv] = l[0];
    }
    double t = 0.0;
    for (int wv = 0; wv < 4; wv++) t += w[wv];
    return t;
}
static double dev_order_sum(const double *term, long n, int nb) {
    double *part = (double *)calloc((size_t)nb, sizeof(double));
    for (int b = 0; b < nb; b++) {
        double v[256];
        for (int t = 0; t < 256; t++) {
            double acc = 0.0;
            for (long i = (long)b * 256 + t; i < n; i += (long)nb * 256) acc += term[i];
            v[t] = acc;
        }
        part[b] = dev_block_sum(v);
    }
    double v[256];
    for (int t = 0; t < 256; t++) {
        double acc = 0.0;
        for (int b = t; b < nb; b += 256) acc += part[b];
        v[t] = acc;
    }
    free(part);
    return dev_block_sum(v);
}
static double sum_terms(const orc_mma_t *M, const double *term, long n) {
    if (M->dev_order > 0) return dev_order_sum(term, n, M->dev_order);
    double sacc = 0.0;
    for (long i = 0; i < n; i++) sacc += term[i];
    return sacc;
}
static double cube(const orc_mma_t *M, double v) { return M->dev_order > 0 ? v * v * v : pow(v, 3.0); }

/* MMA::MMA(n, m, x): a = 0, c = 1000, d = 0 (MMA.cc:108-190); k counts the calls to Update */
ORC_API orc_mma_t *orc_mma_create(long n, int m, const double *x) {
    orc_mma_t *M = (orc_mma_t *)calloc(1, sizeof(orc_mma_t));
    M->n = n;
    M->m = m;
    M->k = 0;
    M->asyminit = 0.5;
    M->asymdec  = 0.7;
    M->asyminc  = 1.2;
    M->a    = (double *)calloc((size_t)m, sizeof(double));
    M->c    = (double *)calloc((size_t)m, sizeof(double));
    M->d    = (double *)calloc((size_t)m, sizeof(double));
    M->y    = (double *)calloc((size_t)m, sizeof(double));
    M->lam  = (double *)calloc((size_t)m, sizeof(double));
    M->mu   = (double *)calloc((size_t)m, sizeof(double));
    M->b    = (double *)calloc((size_t)m, sizeof(double));
    M->grad = (double *)calloc((size_t)m, sizeof(double));
    M->s    = (double *)calloc((size_t)2 * m, sizeof(double));
    M->Hess = (double *)calloc((size_t)m * m, sizeof(double));
    for (int j = 0; j < m; j++) {
        M->a[j] = 0.0;
        M->c[j] = 1000.0;
        M->d[j] = 0.0;
    }
    size_t nb = sizeof(double) * (size_t)n;
    M->L = (double *)calloc(1, nb);
    M->U = (double *)calloc(1, nb);
    M->alpha = (double *)calloc(1, nb);
    M->beta  = (double *)calloc(1, nb);
    M->p0 = (double *)calloc(1, nb);
    M->q0 = (double *)calloc(1, nb);
    M->pij = (double *)calloc((size_t)m, nb);
    M->qij = (double *)calloc((size_t)m, nb);
    M->xo1 = (double *)malloc(nb);
    M->xo2 = (double *)malloc(nb);
    memcpy(M->xo1, x, nb);
    memcpy(M->xo2, x, nb);
    return M;
}
/* nb = number of 256-thread workgroups of the device kernels (csrc/mma.h: grid_for(n, 1024)); 0 = reference order */
ORC_API void orc_mma_set_device_order(orc_mma_t *M, int nb) { M->dev_order = nb; }
ORC_API void orc_mma_destroy(orc_mma_t *M) {
    if (!M) return;
    free(M->a); free(M->c); free(M->d); free(M->y); free(M->lam); free(M->mu); free(M->b); free(M->grad); free(M->s);
    free(M->Hess); free(M->L); free(M->U); free(M->alpha); free(M->beta); free(M->p0); free(M->q0); free(M->pij);
    free(M->qij); free(M->xo1); free(M->xo2);
    free(M);
}
ORC_API void orc_mma_get_state(orc_mma_t *M, double *lam, double *z, double *L, double *U) {
    memcpy(lam, M->lam, sizeof(double) * (size_t)M->m);
    *z = M->z;
    if (L) memcpy(L, M->L, sizeof(double) * (size_t)M->n);
    if (U) memcpy(U, M->U, sizeof(double) * (size_t)M->n);
}

/* SetOuterMovelimit (MMA.cc:386-405) */
ORC_API void orc_mma_outer_movelimit(long n, double Xmin, double Xmax, double movlim, const double *x, double *xmin,
                                     double *xmax) {
    for (long i = 0; i < n; i++) {
        xmax[i] = dmin(Xmax, x[i] + movlim);
        xmin[i] = dmax(Xmin, x[i] - movlim);
    }
}
/* DesignChange (MMA.cc:407-426): inf-norm of x - xold, then xold <- x */
ORC_API double orc_mma_design_change(long n, const double *x, double *xold) {
    double ch = 0.0;
    for (long i = 0; i < n; i++) {
        ch      = fmax(ch, fabs(x[i] - xold[i]));
        xold[i] = x[i];
    }
    return ch;
}

static void gensub(orc_mma_t *M, const double *xv, const double *dfdx, const double *gx, const double *dgdx,
                   const double *xmin, const double *xmax) {
    long n = M->n;
    int m  = M->m;
    M->k++;
    if (M->k < 3) {
        for (long i = 0; i < n; i++) {
            /* VecAXPBYPCZ(L,1,-asyminit,0,xval,xmax); VecAXPY(L,asyminit,xmin)  (MMA.cc:533-536) */
            M->L[i] = (xv[i] + (-M->asyminit) * xmax[i]) + M->asyminit * xmin[i];
            M->U[i] = (xv[i] + M->asyminit * xmax[i]) + (-M->asyminit) * xmin[i];
        }
    } else {
        for (long i = 0; i < n; i++) {
            double helpvar = (xv[i] - M->xo1[i]) * (M->xo1[i] - M->xo2[i]);
            double gamma   = helpvar < 0.0 ? M->asymdec : (helpvar > 0.0 ? M->asyminc : 1.0);
            M->L[i] = xv[i] - gamma * (M->xo1[i] - M->L[i]);
            M->U[i] = xv[i] + gamma * (M->U[i] - M->xo1[i]);
            double xmi = dmax(1.0e-5, xmax[i] - xmin[i]);
            M->L[i] = dmax(M->L[i], xv[i] - 10.0 * xmi);
            M->L[i] = dmin(M->L[i], xv[i] - 0.01 * xmi);
            M->U[i] = dmax(M->U[i], xv[i] + 0.01 * xmi);
            M->U[i] = dmin(M->U[i], xv[i] + 10.0 * xmi);
        }
    }
    const double feps = 1.0e-6;
    for (long i = 0; i < n; i++) {
        double Li = M->L[i], Ui = M->U[i];
        M->alpha[i] = dmax(xmin[i], 0.9 * Li + 0.1 * xv[i]);
        M->beta[i]  = dmin(xmax[i], 0.9 * Ui + 0.1 * xv[i]);
        double dp = dmax(0.0, dfdx[i]), dm = dmax(0.0, -1.0 * dfdx[i]);
        M->p0[i] = pow(Ui - xv[i], 2.0) * (dp + 0.001 * dabs(dfdx[i]) + 0.5 * feps / (Ui - Li));
        M->q0[i] = pow(xv[i] - Li, 2.0) * (dm + 0.001 * dabs(dfdx[i]) + 0.5 * feps / (Ui - Li));
        for (int j = 0; j < m; j++) {
            double g = dgdx[(size_t)j * n + i];
            dp = dmax(0.0, g);
            dm = dmax(0.0, -1.0 * g);
            M->pij[(size_t)j * n + i] = pow(Ui - xv[i], 2.0) * dp; /* constraintModification = false */
            M->qij[(size_t)j * n + i] = pow(xv[i] - Li, 2.0) * dm;
        }
    }
    double *term = (double *)malloc(sizeof(double) * (size_t)n);
    for (int j = 0; j < m; j++) {
        for (long i = 0; i < n; i++)
            term[i] = M->pij[(size_t)j * n + i] / (M->U[i] - xv[i]) + M->qij[(size_t)j * n + i] / (xv[i] - M->L[i]);
        M->b[j] = sum_terms(M, term, n) - gx[j];
    }
    free(term);
}

static void xyz_of_lambda(orc_mma_t *M, double *xv) {
    long n = M->n;
    int m  = M->m;
    double lamai = 0.0;
    for (int j = 0; j < m; j++) {
        if (M->lam[j] < 0.0) M->lam[j] = 0;
        M->y[j] = dmax(0.0, M->lam[j] - M->c[j]);
        lamai += M->lam[j] * M->a[j];
    }
    M->z = dmax(0.0, 10.0 * (lamai - 1.0));
    for (long i = 0; i < n; i++) {
        double pj = M->p0[i], qj = M->q0[i];
        for (int j = 0; j < m; j++) {
            pj += M->pij[(size_t)j * n + i] * M->lam[j];
            qj += M->qij[(size_t)j * n + i] * M->lam[j];
        }
        double v = (sqrt(pj) * M->L[i] + sqrt(qj) * M->U[i]) / (sqrt(pj) + sqrt(qj));
        if (v < M->alpha[i]) v = M->alpha[i];
        if (v > M->beta[i]) v = M->beta[i];
        xv[i] = v;
    }
}

static void dual_grad(orc_mma_t *M, const double *xv) {
    long n = M->n;
    double *term = (double *)malloc(sizeof(double) * (size_t)n);
    for (int j = 0; j < M->m; j++) {
        for (long i = 0; i < n; i++)
            term[i] = M->pij[(size_t)j * n + i] / (M->U[i] - xv[i]) + M->qij[(size_t)j * n + i] / (xv[i] - M->L[i]);
        M->grad[j] = sum_terms(M, term, n) - M->b[j] - M->a[j] * M->z - M->y[j];
    }
    free(term);
}

static void dual_hess(orc_mma_t *M, const double *xv) {
    long n = M->n;
    int m  = M->m;
    double *df2 = (double *)malloc(sizeof(double) * (size_t)n);
    double *PQ  = (double *)malloc(sizeof(double) * (size_t)n * m);
    for (long i = 0; i < n; i++) {
        double pj = M->p0[i], qj = M->q0[i];
        for (int j = 0; j < m; j++) {
            pj += M->pij[(size_t)j * n + i] * M->lam[j];
            qj += M->qij[(size_t)j * n + i] * M->lam[j];
            PQ[i * m + j] = M->pij[(size_t)j * n + i] / pow(M->U[i] - xv[i], 2.0) -
                            M->qij[(size_t)j * n + i] / pow(xv[i] - M->L[i], 2.0);
        }
        df2[i] = -1.0 / (2.0 * pj / cube(M, M->U[i] - xv[i]) + 2.0 * qj / cube(M, xv[i] - M->L[i]));
        double xp = (sqrt(pj) * M->L[i] + sqrt(qj) * M->U[i]) / (sqrt(pj) + sqrt(qj));
        if (xp < M->alpha[i]) df2[i] = 0.0;
        if (xp > M->beta[i]) df2[i] = 0.0;
    }
    for (int i = 0; i < m; i++)
        for (int j = 0; j < m; j++) {
            double *term = (double *)malloc(sizeof(double) * (size_t)n);
            for (long k = 0; k < n; k++) term[k] = (PQ[k * m + i] * df2[k]) * PQ[k * m + j];
            M->Hess[i * m + j] = sum_terms(M, term, n);
            free(term);
        }
    double lamai = 0.0;
    for (int j = 0; j < m; j++) {
        if (M->lam[j] < 0.0) M->lam[j] = 0.0;
        lamai += M->lam[j] * M->a[j];
        if (M->lam[j] > M->c[j]) M->Hess[j * m + j] += -1.0;
        M->Hess[j * m + j] += -M->mu[j] / M->lam[j];
    }
    if (lamai > 0.0)
        for (int j = 0; j < m; j++)
            for (int k = 0; k < m; k++) M->Hess[j * m + k] += -10.0 * M->a[j] * M->a[k];
    double tr = 0.0;
    for (int i = 0; i < m; i++) tr += M->Hess[i * m + i];
    double corr = 1e-4 * tr / m;
    if (-1.0 * corr < 1.0e-7) corr = -1.0e-7;
    for (int i = 0; i < m; i++) M->Hess[i * m + i] += corr;
    free(df2);
    free(PQ);
}

static void lu_factorize(double *K, int nn) {
    for (int ss = 0; ss < nn - 1; ss++)
        for (int i = ss + 1; i < nn; i++) {
            K[i * nn + ss] = K[i * nn + ss] / K[ss * nn + ss];
            for (int j = ss + 1; j < nn; j++) K[i * nn + j] = K[i * nn + j] - K[i * nn + ss] * K[ss * nn + j];
        }
}
static void lu_solve(const double *K, double *x, int nn) {
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

static double dual_residual(orc_mma_t *M, const double *xv, double epsi) {
    long n = M->n;
    int m  = M->m;
    double nrI = 0.0;
    for (int j = 0; j < m; j++) {
        double *term = (double *)malloc(sizeof(double) * (size_t)n);
        for (long i = 0; i < n; i++)
            term[i] = M->pij[(size_t)j * n + i] / (M->U[i] - xv[i]) + M->qij[(size_t)j * n + i] / (xv[i] - M->L[i]);
        const double r = sum_terms(M, term, n);
        free(term);
        double r1 = r - M->b[j] - M->a[j] * M->z - M->y[j] + M->mu[j];
        double r2 = M->mu[j] * M->lam[j] - epsi;
        nrI = dmax(nrI, dmax(dabs(r1), dabs(r2)));
    }
    return nrI;
}

/* MMA::Update (MMA.cc:499-518).  dgdx: m blocks of n.  Returns the number of inner Newton steps. */
ORC_API int orc_mma_update(orc_mma_t *M, double *xval, const double *dfdx, const double *gx, const double *dgdx,
                           const double *xmin, const double *xmax) {
    long n = M->n;
    int m  = M->m;
    gensub(M, xval, dfdx, gx, dgdx, xmin, xmax);
    memcpy(M->xo2, M->xo1, sizeof(double) * (size_t)n);
    memcpy(M->xo1, xval, sizeof(double) * (size_t)n);
    /* SolveDIP (:651-688) */
    for (int j = 0; j < m; j++) {
        M->lam[j] = M->c[j] / 2.0;
        M->mu[j]  = 1.0;
    }
    double tol = 1.0e-9 * sqrt((double)(m + n)), epsi = 1.0, err = 1.0;
    int total = 0;
    while (epsi > tol) {
        int loop = 0;
        while (err > 0.9 * epsi && loop < 100) {
            loop++;
            total++;
            xyz_of_lambda(M, xval);
            dual_grad(M, xval);
            for (int j = 0; j < m; j++) M->grad[j] = -1.0 * M->grad[j] - epsi / M->lam[j];
            dual_hess(M, xval);
            lu_factorize(M->Hess, m);
            lu_solve(M->Hess, M->grad, m);
            for (int j = 0; j < m; j++) M->s[j] = M->grad[j];
            for (int i = 0; i < m; i++) M->s[m + i] = -M->mu[i] + epsi / M->lam[i] - M->s[i] * M->mu[i] / M->lam[i];
            /* DualLineSearch (:882-900) */
            double theta = 1.005;
            for (int i = 0; i < m; i++) {
                if (theta < -1.01 * M->s[i] / M->lam[i]) theta = -1.01 * M->s[i] / M->lam[i];
                if (theta < -1.01 * M->s[i + m] / M->mu[i]) theta = -1.01 * M->s[i + m] / M->mu[i];
            }
            theta = 1.0 / theta;
            for (int i = 0; i < m; i++) {
                M->lam[i] = M->lam[i] + theta * M->s[i];
                M->mu[i]  = M->mu[i] + theta * M->s[i + m];
            }
            xyz_of_lambda(M, xval);
            err = dual_residual(M, xval, epsi);
        }
        epsi = epsi * 0.1;
    }
    return total;
}

/* KKTresidual (MMA.cc:428-496) */
ORC_API void orc_mma_kkt(orc_mma_t *M, const double *x, const double *dfdx, const double *fx, const double *dgdx,
                         const double *xmin, const double *xmax, double *norm2, double *normInf) {
    long n = M->n;
    int m  = M->m;
    double n2 = 0.0, nI = 0.0;
    for (long i = 0; i < n; i++) {
        double ri = dfdx[i];
        for (int j = 0; j < m; j++) ri += M->lam[j] * dgdx[(size_t)j * n + i];
        double mu_min = 0.0, mu_max = 0.0;
        if (x[i] < xmin[i] + 1.0e-5 && ri > 0.0) mu_min = ri;
        if (x[i] > xmax[i] - 1.0e-5 && ri < 0.0) mu_max = -ri;
        ri += -mu_min + mu_max;
        n2 += pow(ri, 2.0);
        nI = dmax(dabs(ri), nI);
        double resi = mu_min * (x[i] - xmin[i]);
        n2 += pow(resi, 2.0);
        nI = dmax(dabs(resi), nI);
        resi = mu_max * (xmax[i] - x[i]);
        n2 += pow(resi, 2.0);
        nI = dmax(dabs(resi), nI);
    }
    double ri = 0.0;
    for (int j = 0; j < m; j++) ri += M->lam[j] * (M->a[j] * M->z + M->y[j] - fx[j]);
    n2 += pow(ri, 2.0);
    nI = dmax(dabs(ri), nI);
    *norm2   = sqrt(n2);
    *normInf = nI;
}

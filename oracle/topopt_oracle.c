/*
 * topopt_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the arithmetic of TopOpt_in_PETSc's hot path
 * (LinearElasticity.cc / Filter.cc / PDEFilter.cc + the PETSc operations they
 * call), written from the reference's behaviour.  It follows the REFERENCE'S
 * data structures -- an assembled sparse matrix K (CSR), explicit
 * interpolation matrices P, Galerkin products P^T A P by sparse
 * matrix-matrix multiplication, an explicit filter matrix H -- which is
 * deliberately a different route from the product's matrix-free HIP kernels,
 * so agreement between the two is evidence, not tautology.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product path never calls it.
 *
 * PARITY PINNING
 *   - orc_hex8_ke / orc_pde_kf are pinned against vectors produced by running
 *     the reference's own PETSc-free functions (tests/golden/ref_*.bin, made by
 *     tests/golden/make_ref_vectors.sh) and against the known answers recorded
 *     in SURVEY.md section 8(a).
 *   - Everything that the reference delegates to PETSc 3.11 (KSPCG, PCMG,
 *     Chebyshev, DMDA Q1 interpolation, MatPtAP) is restated from the published
 *     algorithms; PETSc is not installable here, the reference ships no tests:
 *     for those parts PARITY IS UNPINNED against the reference and is anchored
 *     instead by solver-independent invariants (tests/test_oracle_*.py).
 *
 * THREADS.  The loops that carry the time of a design iteration (CSR assembly, Galerkin SpGEMM, CSR products, restriction,
 * vector updates, compliance) run under OpenMP, and every one of them is written so that a thread computes WHOLE result
 * entries with the serial form's order of additions (reductions: fixed chunks, partial sums added in chunk order): results do
 * not depend on the thread count, bit for bit (tests: 1 against 8 threads).  bench.py's cpu_baseline uses all usable cores.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp -shared).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* small helpers                                                             */
/* ------------------------------------------------------------------------- */

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) {
        fprintf(stderr, "topopt_oracle: out of memory (%zu bytes)\n", n);
        abort();
    }
    return p;
}
static void *xcalloc(size_t n, size_t s) {
    void *p = calloc(n ? n : 1, s ? s : 1);
    if (!p) {
        fprintf(stderr, "topopt_oracle: out of memory\n");
        abort();
    }
    return p;
}

/* splitmix64 -> uniform double in [0,1).  Shared definition of the synthetic,
 * partition-independent random fields; the product has its
 * own copy of this 6-line integer hash. */
static inline double hash_u01(uint64_t idx, uint64_t seed) {
    uint64_t z = (idx + 1u) * 0x9E3779B97F4A7C15ULL + seed;
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
ORC_API double orc_hash_u01(uint64_t idx, uint64_t seed) { return hash_u01(idx, seed); }

/* deterministic (thread-count independent) blocked reductions */
#define RED_CHUNK 4096
/* element-wise loops over long vectors: OpenMP where the vector is long enough to pay for the fork (every thread
 * computes whole entries: results do not depend on the thread count) */
#define PFOR_MIN 100000
static double vdot(long n, const double *a, const double *b) {
    long nch = (n + RED_CHUNK - 1) / RED_CHUNK;
    double *part = (double *)xmalloc(sizeof(double) * (size_t)nch);
#pragma omp parallel for schedule(static) if (n > 400000)
    for (long c = 0; c < nch; c++) {
        long lo = c * RED_CHUNK, hi = lo + RED_CHUNK > n ? n : lo + RED_CHUNK;
        double s = 0.0;
        for (long i = lo; i < hi; i++) s += a[i] * b[i];
        part[c] = s;
    }
    double s = 0.0;
    for (long c = 0; c < nch; c++) s += part[c];
    free(part);
    return s;
}
static double vnorm(long n, const double *a) { return sqrt(vdot(n, a, a)); }
static double vsum(long n, const double *a) {
    long nch = (n + RED_CHUNK - 1) / RED_CHUNK;
    double *part = (double *)xmalloc(sizeof(double) * (size_t)nch);
#pragma omp parallel for schedule(static) if (n > 400000)
    for (long c = 0; c < nch; c++) {
        long lo = c * RED_CHUNK, hi = lo + RED_CHUNK > n ? n : lo + RED_CHUNK;
        double s = 0.0;
        for (long i = lo; i < hi; i++) s += a[i];
        part[c] = s;
    }
    double s = 0.0;
    for (long c = 0; c < nch; c++) s += part[c];
    free(part);
    return s;
}

/* ------------------------------------------------------------------------- */
/* a1. Hex8 element stiffness  (LinearElasticity.cc:841-1057)                 */
/* ------------------------------------------------------------------------- */

static double dot8(const double *a, const double *b) { /* LinearElasticity.cc:999-1007 */
    double r = 0.0;
    for (int i = 0; i < 8; i++) r = r + a[i] * b[i];
    return r;
}

/* LinearElasticity.cc:1009-1040: derivatives of the trilinear shape functions;
 * node order: counter-clockwise in the z=-1 plane, then the z=+1 plane. */
static void dshape(double xi, double eta, double zeta, double *dxi, double *deta, double *dzeta) {
    static const double sx[8] = {-1, 1, 1, -1, -1, 1, 1, -1};
    static const double sy[8] = {-1, -1, 1, 1, -1, -1, 1, 1};
    static const double sz[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
    for (int a = 0; a < 8; a++) {
        /* written as (+-0.125)*(1 -+ eta)*(1 -+ zeta) exactly like the reference:
         * (1 + s*eta) with s=-1 is (1 - eta) bit-for-bit */
        dxi[a]   = (sx[a] * 0.125) * (1.0 + sy[a] * eta) * (1.0 + sz[a] * zeta);
        deta[a]  = (sy[a] * 0.125) * (1.0 + sx[a] * xi) * (1.0 + sz[a] * zeta);
        dzeta[a] = (sz[a] * 0.125) * (1.0 + sx[a] * xi) * (1.0 + sy[a] * eta);
    }
}

static double inv3(double J[3][3], double iJ[3][3]) { /* LinearElasticity.cc:1042-1057 */
    double det = J[0][0] * (J[1][1] * J[2][2] - J[2][1] * J[1][2]) - J[0][1] * (J[1][0] * J[2][2] - J[2][0] * J[1][2]) +
                 J[0][2] * (J[1][0] * J[2][1] - J[2][0] * J[1][1]);
    iJ[0][0] = (J[1][1] * J[2][2] - J[2][1] * J[1][2]) / det;
    iJ[0][1] = -(J[0][1] * J[2][2] - J[0][2] * J[2][1]) / det;
    iJ[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
    iJ[1][0] = -(J[1][0] * J[2][2] - J[1][2] * J[2][0]) / det;
    iJ[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
    iJ[1][2] = -(J[0][0] * J[1][2] - J[0][2] * J[1][0]) / det;
    iJ[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) / det;
    iJ[2][1] = -(J[0][0] * J[2][1] - J[0][1] * J[2][0]) / det;
    iJ[2][2] = (J[0][0] * J[1][1] - J[1][0] * J[0][1]) / det;
    return det;
}

/* KE = sum_GP w detJ B^T C B with E = 1 (LinearElasticity.cc:885-996).  The
 * accumulation order (ii,jj,kk, then i,j,k,l) is the reference's, because KE is
 * symmetric only to ~1e-17 and bit-level fixtures depend on the order. */
ORC_API void orc_hex8_ke(const double *X, const double *Y, const double *Z, double nu, int redInt, double *ke) {
    double lambda = nu / ((1.0 + nu) * (1.0 - 2.0 * nu));
    double mu     = 1.0 / (2.0 * (1.0 + nu));
    double C[6][6];
    memset(C, 0, sizeof(C));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[i][j] = (i == j) ? lambda + 2.0 * mu : lambda;
    C[3][3] = C[4][4] = C[5][5] = mu;
    double GP[2] = {-0.577350269189626, 0.577350269189626};
    double W[2]  = {1.0, 1.0};
    if (redInt) {
        GP[0] = 0.0;
        W[0]  = 2.0;
    }
    /* strain pattern: which displacement component (col) feeds strain row,
     * for derivative direction 1,2,3 (alpha1..3 at LinearElasticity.cc:907-921) */
    double al[3][6][3];
    memset(al, 0, sizeof(al));
    al[0][0][0] = 1.0; al[0][3][1] = 1.0; al[0][5][2] = 1.0;
    al[1][1][1] = 1.0; al[1][3][0] = 1.0; al[1][4][2] = 1.0;
    al[2][2][2] = 1.0; al[2][4][1] = 1.0; al[2][5][0] = 1.0;
    double dN[3][8], J[3][3], iJ[3][3], beta[6][3], B[6][24];
    memset(ke, 0, sizeof(double) * 576);
    int ng = 2 - (redInt ? 1 : 0);
    for (int ii = 0; ii < ng; ii++)
        for (int jj = 0; jj < ng; jj++)
            for (int kk = 0; kk < ng; kk++) {
                dshape(GP[ii], GP[jj], GP[kk], dN[0], dN[1], dN[2]);
                for (int r = 0; r < 3; r++) {
                    J[r][0] = dot8(dN[r], X);
                    J[r][1] = dot8(dN[r], Y);
                    J[r][2] = dot8(dN[r], Z);
                }
                double detJ   = inv3(J, iJ);
                double weight = W[ii] * W[jj] * W[kk] * detJ;
                memset(B, 0, sizeof(B));
                for (int ll = 0; ll < 3; ll++) {
                    for (int i = 0; i < 6; i++)
                        for (int j = 0; j < 3; j++)
                            beta[i][j] = iJ[0][ll] * al[0][i][j] + iJ[1][ll] * al[1][i][j] + iJ[2][ll] * al[2][i][j];
                    for (int i = 0; i < 6; i++)
                        for (int j = 0; j < 24; j++) B[i][j] = B[i][j] + beta[i][j % 3] * dN[ll][j / 3];
                }
                for (int i = 0; i < 24; i++)
                    for (int j = 0; j < 24; j++)
                        for (int k = 0; k < 6; k++)
                            for (int l = 0; l < 6; l++)
                                ke[j + 24 * i] = ke[j + 24 * i] + weight * (B[k][i] * C[k][l] * B[l][j]);
            }
}

/* box element of size dx,dy,dz as set up at LinearElasticity.cc:118-123 */
ORC_API void orc_hex8_ke_box(double dx, double dy, double dz, double nu, double *ke) {
    double X[8] = {0.0, dx, dx, 0.0, 0.0, dx, dx, 0.0};
    double Y[8] = {0.0, 0.0, dy, dy, 0.0, 0.0, dy, dy};
    double Z[8] = {0.0, 0.0, 0.0, 0.0, dz, dz, dz, dz};
    orc_hex8_ke(X, Y, Z, nu, 0, ke);
}

/* ------------------------------------------------------------------------- */
/* a12. Helmholtz filter element matrices  (PDEFilter.cc:472-576)             */
/* KF = R^2 int gradN.gradN + int N N in closed form, TF = 1/8.               */
/* ------------------------------------------------------------------------- */
ORC_API void orc_pde_kf(double dx, double dy, double dz, double RR, double *KK, double *T) {
    double a   = 1.0 / dx / dy;
    double b   = 1 / dz;
    double r2  = RR * RR;
    double x2  = dx * dx, y2 = dy * dy, z2 = dz * dz;
    double rx  = r2 * x2;
    double rxy = rx * y2;  /* R^2 dx^2 dy^2 */
    double rxz = rx * z2;  /* R^2 dx^2 dz^2 */
    double ryz = r2 * y2 * z2;
    double m   = x2 * y2 * z2;
    double A3 = 3.0 * rxy, B3 = 3.0 * rxz, C3 = 3.0 * ryz;
    double A6 = 6.0 * rxy, B6 = 6.0 * rxz, C6 = 6.0 * ryz;
    /* the eight distinct entries, by relative node position (PDEFilter.cc:490-500) */
    double k_self = a * b * (A3 + B3 + C3 + m) / 27.0;  /* same node           */
    double k_x    = a * b * (A3 + B3 - C6 + m) / 54.0;  /* neighbour along x   */
    double k_xy   = a * b * (A3 - B6 - C6 + m) / 108.0; /* diagonal in xy      */
    double k_y    = a * b * (A3 - B6 + C3 + m) / 54.0;  /* neighbour along y   */
    double k_z    = -(a * b * (A6 - B3 - C3 - m) / 54.0);
    double k_xz   = -(a * b * (A6 - B3 + C6 - m) / 108.0);
    double k_xyz  = -(a * b * (A6 + B6 + C6 - m) / 216.0);
    double k_yz   = -(a * b * (A6 + B6 - C3 - m) / 108.0);
    static const int lx[8] = {0, 1, 1, 0, 0, 1, 1, 0}, ly[8] = {0, 0, 1, 1, 0, 0, 1, 1}, lz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
    for (int p = 0; p < 8; p++)
        for (int q = 0; q < 8; q++) {
            int ddx = lx[p] != lx[q], ddy = ly[p] != ly[q], ddz = lz[p] != lz[q];
            double v;
            if (!ddz) v = !ddx ? (!ddy ? k_self : k_y) : (!ddy ? k_x : k_xy);
            else v = !ddx ? (!ddy ? k_z : k_yz) : (!ddy ? k_xz : k_xyz);
            KK[8 * p + q] = v;
        }
    for (int p = 0; p < 8; p++) T[p] = 0.125 * 1.0;
}

/* ------------------------------------------------------------------------- */
/* a2. structured hex mesh bookkeeping (LinearElasticity.cc:785-839, serial)  */
/* node id = i + nx*(j + ny*k); element id = i + ex*(j + ey*k);               */
/* local node order (:819-826): (0,0,0)(1,0,0)(1,1,0)(0,1,0) then z+1.        */
/* ------------------------------------------------------------------------- */
static const int LX[8] = {0, 1, 1, 0, 0, 1, 1, 0};
static const int LY[8] = {0, 0, 1, 1, 0, 0, 1, 1};
static const int LZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};

static inline void elem_nodes(int nx, int ny, int i, int j, int k, long nd[8]) {
    for (int a = 0; a < 8; a++) nd[a] = (long)(i + LX[a]) + (long)nx * ((long)(j + LY[a]) + (long)ny * (k + LZ[a]));
}

/* a3. cantilever load & Dirichlet vectors (LinearElasticity.cc:143-171).
 * Coordinates are xmin + i*h (DMDASetUniformCoordinates), tests use the same
 * epsilon window as the reference. */
ORC_API void orc_cantilever_bc(int nx, int ny, int nz, double hx, double hy, double hz, double *N, double *RHS) {
    double xc[6] = {0.0, (nx - 1) * hx, 0.0, (ny - 1) * hy, 0.0, (nz - 1) * hz};
    double eps   = fmin(hx * 0.05, fmin(hy * 0.05, hz * 0.05));
    long nn      = (long)nx * ny * nz;
    for (long n = 0; n < 3 * nn; n++) {
        N[n]   = 1.0;
        RHS[n] = 0.0;
    }
    for (int k = 0; k < nz; k++)
        for (int j = 0; j < ny; j++)
            for (int i = 0; i < nx; i++) {
                long n   = (long)i + (long)nx * (j + (long)ny * k);
                double x = xc[0] + i * hx, y = xc[2] + j * hy, z = xc[4] + k * hz;
                if (fabs(x - xc[0]) < eps) N[3 * n] = N[3 * n + 1] = N[3 * n + 2] = 0.0;
                if (fabs(x - xc[1]) < eps && fabs(z - xc[4]) < eps) RHS[3 * n + 2] = -0.001;
                if (fabs(x - xc[1]) < eps && fabs(y - xc[2]) < eps && fabs(z - xc[4]) < eps) RHS[3 * n + 2] = -0.001 / 2.0;
                if (fabs(x - xc[1]) < eps && fabs(y - xc[3]) < eps && fabs(z - xc[4]) < eps) RHS[3 * n + 2] = -0.001 / 2.0;
            }
}

/* SIMP interpolation E = Emin + x^p (Emax-Emin)  (LinearElasticity.cc:519) */
ORC_API void orc_simp(long n, const double *x, double Emin, double Emax, double penal, double *E) {
    for (long e = 0; e < n; e++) E[e] = Emin + pow(x[e], penal) * (Emax - Emin);
}

/* Synthetic "mid-optimisation" density of SURVEY.md 8(d):
 * clamp(0.12 + 0.4 sin(7 pi x) sin(5 pi y) sin(3 pi z) + 0.3 (u-0.5), 1e-3, 1),
 * u from a 64-bit hash of the GLOBAL element id (partition independent).
 * e0z = first global element layer of this block (0 for the whole mesh). */
ORC_API void orc_synth_density(int ex, int ey, int ez, int e0z, int ez_glob, double h, uint64_t seed, double *x) {
    (void)ez_glob;
    const double pi = 3.14159265358979323846;
    for (int k = 0; k < ez; k++)
        for (int j = 0; j < ey; j++)
            for (int i = 0; i < ex; i++) {
                uint64_t gid = (uint64_t)i + (uint64_t)ex * ((uint64_t)j + (uint64_t)ey * (uint64_t)(k + e0z));
                double xc = (i + 0.5) * h, yc = (j + 0.5) * h, zc = (k + e0z + 0.5) * h;
                double v  = 0.12 + 0.4 * sin(7 * pi * xc) * sin(5 * pi * yc) * sin(3 * pi * zc) +
                           0.3 * (hash_u01(gid, seed) - 0.5);
                if (v < 1e-3) v = 1e-3;
                if (v > 1.0) v = 1.0;
                x[(long)i + (long)ex * (j + (long)ey * k)] = v;
            }
}

/* ------------------------------------------------------------------------- */
/* matrix-free apply  y = (N K(E) N + I - N) u  (the operator the reference   */
/* assembles at LinearElasticity.cc:510-542), element-by-element scatter.     */
/* dof = 3 (elasticity, KE 24x24) or 1 (Helmholtz, KF 8x8). E, N may be NULL. */
/* ------------------------------------------------------------------------- */
ORC_API void orc_matfree_apply(int nx, int ny, int nz, int dof, const double *KE, const double *E, const double *N,
                               const double *u, double *y) {
    int ex = nx - 1, ey = ny - 1, ez = nz - 1, ed = 8 * dof;
    long nn = (long)nx * ny * nz;
    for (long n = 0; n < dof * nn; n++) y[n] = 0.0;
    double ue[24], fe[24];
    long nd[8];
    for (int k = 0; k < ez; k++)
        for (int j = 0; j < ey; j++)
            for (int i = 0; i < ex; i++) {
                long e = (long)i + (long)ex * (j + (long)ey * k);
                elem_nodes(nx, ny, i, j, k, nd);
                for (int a = 0; a < 8; a++)
                    for (int c = 0; c < dof; c++) {
                        long g         = dof * nd[a] + c;
                        ue[dof * a + c] = N ? N[g] * u[g] : u[g];
                    }
                double s = E ? E[e] : 1.0;
                for (int r = 0; r < ed; r++) {
                    double acc = 0.0;
                    for (int c = 0; c < ed; c++) acc += KE[r * ed + c] * ue[c];
                    fe[r] = s * acc;
                }
                for (int a = 0; a < 8; a++)
                    for (int c = 0; c < dof; c++) y[dof * nd[a] + c] += fe[dof * a + c];
            }
    if (N)
        for (long n = 0; n < dof * nn; n++) y[n] = N[n] * y[n] + (1.0 - N[n]) * u[n];
}

/* ------------------------------------------------------------------------- */
/* CSR matrices                                                              */
/* ------------------------------------------------------------------------- */
typedef struct {
    long nrow, ncol;
    long *rp;   /* nrow+1 */
    int *ci;    /* nnz */
    double *v;  /* nnz */
} csr_t;

static void csr_free(csr_t *A) {
    if (!A) return;
    free(A->rp);
    free(A->ci);
    free(A->v);
    free(A);
}

/* CPU baseline variant (bench.py, SURVEY 8d "assembled-CSR and matrix-free variants"): the fine-level operator of the
 * solve applied from KE and the moduli, one node per loop trip (gather over the 8 adjacent elements: no write
 * conflicts between threads), instead of the assembled CSR.  Same operator (N K N + I - N), other summation order. */
static const csr_t *g_mf_A = NULL; /* the matrix the hook stands in for */
static const double *g_mf_KE, *g_mf_E, *g_mf_N;
static int g_mf_nx, g_mf_ny, g_mf_nz;
static void matfree_apply_omp(const double *u, double *y) {
    const int nx = g_mf_nx, ny = g_mf_ny, nz = g_mf_nz, ex = nx - 1, ey = ny - 1, ez = nz - 1;
    const double *KE = g_mf_KE, *E = g_mf_E, *N = g_mf_N;
    static const int LX[8] = {0, 1, 1, 0, 0, 1, 1, 0}, LY[8] = {0, 0, 1, 1, 0, 0, 1, 1}, LZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < nz; k++)
        for (int j = 0; j < ny; j++)
            for (int i = 0; i < nx; i++) {
                const long n = (long)i + (long)nx * (j + (long)ny * k);
                double acc[3] = {0.0, 0.0, 0.0};
                for (int a = 0; a < 8; a++) { /* the element in which this node is corner a */
                    const int ei = i - LX[a], ej = j - LY[a], ek = k - LZ[a];
                    if (ei < 0 || ei >= ex || ej < 0 || ej >= ey || ek < 0 || ek >= ez) continue;
                    const double Ee = E ? E[(long)ei + (long)ex * (ej + (long)ey * ek)] : 1.0;
                    double f[3] = {0.0, 0.0, 0.0};
                    for (int b = 0; b < 8; b++) {
                        const long nb = (long)(ei + LX[b]) + (long)nx * ((ej + LY[b]) + (long)ny * (ek + LZ[b]));
                        for (int c = 0; c < 3; c++) {
                            const double ub = N ? N[3 * nb + c] * u[3 * nb + c] : u[3 * nb + c];
                            for (int r = 0; r < 3; r++) f[r] += KE[(3 * a + r) * 24 + 3 * b + c] * ub;
                        }
                    }
                    for (int r = 0; r < 3; r++) acc[r] += Ee * f[r];
                }
                for (int r = 0; r < 3; r++) y[3 * n + r] = N ? N[3 * n + r] * acc[r] + (1.0 - N[3 * n + r]) * u[3 * n + r] : acc[r];
            }
}

static void csr_spmv(const csr_t *A, const double *x, double *y) {
    if (A == g_mf_A && A) {
        matfree_apply_omp(x, y);
        return;
    }
#pragma omp parallel for schedule(static) if (A->nrow > 100000)
    for (long r = 0; r < A->nrow; r++) {
        double s = 0.0;
        for (long p = A->rp[r]; p < A->rp[r + 1]; p++) s += A->v[p] * x[A->ci[p]];
        y[r] = s;
    }
}
/* y = A^T x (serial scatter; used for T^T and where no transpose is stored.  The multigrid restriction uses the stored
 * transpose -- csr_transpose keeps the rows of a column in ascending order, so the gather adds the same terms in the
 * same order as this scatter: same bits, any number of threads) */
static void csr_spmv_t(const csr_t *A, const double *x, double *y) {
    for (long c = 0; c < A->ncol; c++) y[c] = 0.0;
    for (long r = 0; r < A->nrow; r++)
        for (long p = A->rp[r]; p < A->rp[r + 1]; p++) y[A->ci[p]] += A->v[p] * x[r];
}

static csr_t *csr_transpose(const csr_t *A) {
    csr_t *T = (csr_t *)xcalloc(1, sizeof(csr_t));
    long nnz = A->rp[A->nrow];
    T->nrow  = A->ncol;
    T->ncol  = A->nrow;
    T->rp    = (long *)xcalloc((size_t)T->nrow + 1, sizeof(long));
    T->ci    = (int *)xmalloc(sizeof(int) * (size_t)nnz);
    T->v     = (double *)xmalloc(sizeof(double) * (size_t)nnz);
    for (long p = 0; p < nnz; p++) T->rp[A->ci[p] + 1]++;
    for (long r = 0; r < T->nrow; r++) T->rp[r + 1] += T->rp[r];
    long *pos = (long *)xmalloc(sizeof(long) * (size_t)T->nrow);
    memcpy(pos, T->rp, sizeof(long) * (size_t)T->nrow);
    for (long r = 0; r < A->nrow; r++)
        for (long p = A->rp[r]; p < A->rp[r + 1]; p++) {
            long q   = pos[A->ci[p]]++;
            T->ci[q] = (int)r;
            T->v[q]  = A->v[p];
        }
    free(pos);
    return T;
}

typedef struct {
    int c;
    double v;
} cv_t;
static int cv_cmp(const void *a, const void *b) { return ((const cv_t *)a)->c - ((const cv_t *)b)->c; }

/* C = A * B, Gustavson SpGEMM with a dense accumulator per thread; columns sorted.  Two passes over the rows (count,
 * then fill), both OpenMP-parallel over rows: a row of C is computed by one thread with the same accumulation order as
 * the serial form, so the result does not depend on the thread count. */
static csr_t *csr_matmul(const csr_t *A, const csr_t *B) {
    csr_t *C = (csr_t *)xcalloc(1, sizeof(csr_t));
    C->nrow  = A->nrow;
    C->ncol  = B->ncol;
    C->rp    = (long *)xcalloc((size_t)C->nrow + 1, sizeof(long));
    const int par = A->nrow > 20000;
    /* pass 1: row lengths */
#pragma omp parallel if (par)
    {
        long *mark = (long *)xmalloc(sizeof(long) * (size_t)B->ncol);
        for (long c = 0; c < B->ncol; c++) mark[c] = -1;
#pragma omp for schedule(dynamic, 512)
        for (long r = 0; r < A->nrow; r++) {
            long cnt = 0;
            for (long p = A->rp[r]; p < A->rp[r + 1]; p++) {
                int k = A->ci[p];
                for (long q = B->rp[k]; q < B->rp[k + 1]; q++) {
                    int c = B->ci[q];
                    if (mark[c] != r) {
                        mark[c] = r;
                        cnt++;
                    }
                }
            }
            C->rp[r + 1] = cnt;
        }
        free(mark);
    }
    for (long r = 0; r < C->nrow; r++) C->rp[r + 1] += C->rp[r];
    long nnz = C->rp[C->nrow];
    C->ci    = (int *)xmalloc(sizeof(int) * (size_t)(nnz + 1));
    C->v     = (double *)xmalloc(sizeof(double) * (size_t)(nnz + 1));
    /* pass 2: values */
#pragma omp parallel if (par)
    {
        double *acc = (double *)xcalloc((size_t)B->ncol, sizeof(double));
        long *mark  = (long *)xmalloc(sizeof(long) * (size_t)B->ncol);
        for (long c = 0; c < B->ncol; c++) mark[c] = -1;
        cv_t *tmp   = NULL;
        long tmpcap = 0;
#pragma omp for schedule(dynamic, 512)
        for (long r = 0; r < A->nrow; r++) {
            const long cnt_r = C->rp[r + 1] - C->rp[r];
            if (cnt_r > tmpcap) {
                tmpcap = cnt_r * 2;
                tmp    = (cv_t *)realloc(tmp, sizeof(cv_t) * (size_t)tmpcap);
                if (!tmp) abort();
            }
            long cnt = 0;
            for (long p = A->rp[r]; p < A->rp[r + 1]; p++) {
                int k     = A->ci[p];
                double av = A->v[p];
                for (long q = B->rp[k]; q < B->rp[k + 1]; q++) {
                    int c = B->ci[q];
                    if (mark[c] != r) {
                        mark[c]      = r;
                        acc[c]       = 0.0;
                        tmp[cnt++].c = c;
                    }
                    acc[c] += av * B->v[q];
                }
            }
            for (long t = 0; t < cnt; t++) tmp[t].v = acc[tmp[t].c];
            qsort(tmp, (size_t)cnt, sizeof(cv_t), cv_cmp);
            long o = C->rp[r];
            for (long t = 0; t < cnt; t++) {
                C->ci[o + t] = tmp[t].c;
                C->v[o + t]  = tmp[t].v;
            }
        }
        free(acc);
        free(mark);
        free(tmp);
    }
    return C;
}

/* a4. Assembled operator as the reference builds it
 * (LinearElasticity.cc:503-542, PDEFilter.cc:243-267):
 *   K = sum_e E_e KE  (MatSetValuesLocal ADD_VALUES), then
 *   K <- N K N (MatDiagonalScale), K <- K + (I - N) (MatDiagonalSet ADD). */
static csr_t *assemble_csr(int nx, int ny, int nz, int dof, const double *KE, const double *E, const double *N) {
    int ex = nx - 1, ey = ny - 1, ez = nz - 1, ed = 8 * dof;
    long nn  = (long)nx * ny * nz;
    csr_t *A = (csr_t *)xcalloc(1, sizeof(csr_t));
    A->nrow = A->ncol = dof * nn;
    A->rp             = (long *)xcalloc((size_t)A->nrow + 1, sizeof(long));
    /* row lengths: (#valid neighbours) * dof */
    for (int k = 0; k < nz; k++)
        for (int j = 0; j < ny; j++)
            for (int i = 0; i < nx; i++) {
                int cx = 1 + (i > 0) + (i < nx - 1), cy = 1 + (j > 0) + (j < ny - 1), cz = 1 + (k > 0) + (k < nz - 1);
                long n = (long)i + (long)nx * (j + (long)ny * k);
                for (int c = 0; c < dof; c++) A->rp[dof * n + c + 1] = (long)cx * cy * cz * dof;
            }
    for (long r = 0; r < A->nrow; r++) A->rp[r + 1] += A->rp[r];
    long nnz = A->rp[A->nrow];
    A->ci    = (int *)xmalloc(sizeof(int) * (size_t)nnz);
    A->v     = (double *)xcalloc((size_t)nnz, sizeof(double));
    /* column pattern, ascending */
#pragma omp parallel for schedule(static) if (nn > 20000)
    for (int k = 0; k < nz; k++)
        for (int j = 0; j < ny; j++)
            for (int i = 0; i < nx; i++) {
                long n = (long)i + (long)nx * (j + (long)ny * k);
                for (int c = 0; c < dof; c++) {
                    long p = A->rp[dof * n + c];
                    for (int dk = -1; dk <= 1; dk++) {
                        if (k + dk < 0 || k + dk >= nz) continue;
                        for (int dj = -1; dj <= 1; dj++) {
                            if (j + dj < 0 || j + dj >= ny) continue;
                            for (int di = -1; di <= 1; di++) {
                                if (i + di < 0 || i + di >= nx) continue;
                                long m = (long)(i + di) + (long)nx * ((j + dj) + (long)ny * (k + dk));
                                for (int cc = 0; cc < dof; cc++) A->ci[p++] = (int)(dof * m + cc);
                            }
                        }
                    }
                }
            }
    /* values, gathered per NODE (OpenMP over nodes, no write conflicts): node n collects, from each of its up to 8
     * elements in ASCENDING element order, the rows of the element matrix that belong to it.  The reference's element
     * loop (LinearElasticity.cc:503-528) adds to an entry once per element in ascending order too, so every entry sees
     * the same additions in the same order: bit-identical to the serial element loop, for any thread count. */
#pragma omp parallel for collapse(2) schedule(static) if (nn > 20000)
    for (int k = 0; k < nz; k++)
        for (int j = 0; j < ny; j++)
            for (int i = 0; i < nx; i++) {
                const long n = (long)i + (long)nx * (j + (long)ny * k);
                const int cx = 1 + (i > 0) + (i < nx - 1), cy = 1 + (j > 0) + (j < ny - 1);
                for (int dk = -1; dk <= 0; dk++)
                    for (int dj = -1; dj <= 0; dj++)
                        for (int di = -1; di <= 0; di++) { /* element (i + di, j + dj, k + dk): ascending e */
                            const int ei = i + di, ej = j + dj, ek = k + dk;
                            if (ei < 0 || ei >= ex || ej < 0 || ej >= ey || ek < 0 || ek >= ez) continue;
                            const long e   = (long)ei + (long)ex * (ej + (long)ey * ek);
                            const double s = E ? E[e] : 1.0;
                            int a = -1; /* local number of node n in that element */
                            for (int q = 0; q < 8; q++)
                                if (LX[q] == -di && LY[q] == -dj && LZ[q] == -dk) a = q;
                            for (int b = 0; b < 8; b++) {
                                const int ddi = LX[b] - LX[a], ddj = LY[b] - LY[a], ddk = LZ[b] - LZ[a];
                                /* slot of neighbour (ddi,ddj,ddk) in node n's sorted neighbour list */
                                const int sx = ddi + (i > 0), sy = ddj + (j > 0), sz = ddk + (k > 0);
                                const long slot = ((long)sz * cy + sy) * cx + sx;
                                for (int c = 0; c < dof; c++) {
                                    const long p = A->rp[dof * n + c] + slot * dof;
                                    for (int cc = 0; cc < dof; cc++) A->v[p + cc] += KE[(dof * a + c) * ed + dof * b + cc] * s;
                                }
                            }
                        }
            }
    if (N) {
#pragma omp parallel for schedule(static) if (A->nrow > PFOR_MIN)
        for (long r = 0; r < A->nrow; r++)
            for (long p = A->rp[r]; p < A->rp[r + 1]; p++) {
                A->v[p] = N[r] * A->v[p] * N[A->ci[p]];
                if (A->ci[p] == r) A->v[p] += 1.0 - N[r];
            }
    }
    return A;
}

/* Trilinear (Q1) node interpolation coarse -> fine, factor-2 coarsening with
 * coarse node I at fine node 2I: what DMCreateInterpolation(DMDA Q1) builds
 * (LinearElasticity.cc:704, PDEFilter.cc:332). */
static csr_t *interp_csr(int ncx, int ncy, int ncz, int dof) {
    int nfx = 2 * (ncx - 1) + 1, nfy = 2 * (ncy - 1) + 1, nfz = 2 * (ncz - 1) + 1;
    long nf = (long)nfx * nfy * nfz, nc = (long)ncx * ncy * ncz;
    csr_t *P = (csr_t *)xcalloc(1, sizeof(csr_t));
    P->nrow  = dof * nf;
    P->ncol  = dof * nc;
    P->rp    = (long *)xcalloc((size_t)P->nrow + 1, sizeof(long));
    P->ci    = (int *)xmalloc(sizeof(int) * (size_t)(8 * P->nrow));
    P->v     = (double *)xmalloc(sizeof(double) * (size_t)(8 * P->nrow));
    long nnz = 0;
    for (int k = 0; k < nfz; k++)
        for (int j = 0; j < nfy; j++)
            for (int i = 0; i < nfx; i++) {
                int ci[2] = {i / 2, i / 2 + 1}, cj[2] = {j / 2, j / 2 + 1}, ck[2] = {k / 2, k / 2 + 1};
                int mi = (i & 1) ? 2 : 1, mj = (j & 1) ? 2 : 1, mk = (k & 1) ? 2 : 1;
                double wi = (i & 1) ? 0.5 : 1.0, wj = (j & 1) ? 0.5 : 1.0, wk = (k & 1) ? 0.5 : 1.0;
                long n = (long)i + (long)nfx * (j + (long)nfy * k);
                for (int c = 0; c < dof; c++) {
                    for (int kk = 0; kk < mk; kk++)
                        for (int jj = 0; jj < mj; jj++)
                            for (int ii = 0; ii < mi; ii++) {
                                long m     = (long)ci[ii] + (long)ncx * (cj[jj] + (long)ncy * ck[kk]);
                                P->ci[nnz] = (int)(dof * m + c);
                                P->v[nnz]  = wi * wj * wk;
                                nnz++;
                            }
                    P->rp[dof * n + c + 1] = nnz;
                }
            }
    return P;
}

/* ------------------------------------------------------------------------- */
/* a5/a6/a15. Multigrid-preconditioned CG on the assembled hierarchy.         */
/*  outer  : KSPCG, unpreconditioned residual norm, reference norm ||b||      */
/*           (KSPConvergedDefault with a nonzero initial guess),             */
/*  PC     : PCMG multiplicative V-cycle, Galerkin coarse operators P^T A P,  */
/*  levels : Chebyshev(k) + Jacobi with eigenvalue window [lo,hi]*lam_est,    */
/*  coarse : Chebyshev(k_coarse) + Jacobi.                                    */
/* lam_est : level 0 -- the element bound  max eig(diag(KE)^-1 KE) (rigorous  */
/*           for sum-of-element operators, density independent);             */
/*           level>0 -- largest Ritz value of a 10-step Lanczos run on        */
/*           D^-1/2 A D^-1/2 from a fixed hashed start vector.                */
/* ------------------------------------------------------------------------- */
#define MAXLV 12
typedef struct {
    int nlv, dof;
    int nx[MAXLV], ny[MAXLV], nz[MAXLV];
    csr_t *A[MAXLV];
    csr_t *P[MAXLV]; /* P[l]: level l+1 (coarse) -> level l (fine) */
    csr_t *PT[MAXLV]; /* its transpose, stored: restriction as a gather, Galerkin products */
    double *dinv[MAXLV];
    double lam[MAXLV];
    double lam_min[MAXLV]; /* coarsest level only: smallest Ritz value (coarse-solve window) */
    double *b[MAXLV], *x[MAXLV], *r[MAXLV], *d[MAXLV];
    int nsmooth, ncoarse, nlanczos, nlanczos_coarse, fine_eig;
    double cheb_lo, cheb_hi;
    int cycles[MAXLV]; /* cycles[l]: how often level l + 1 is cycled from level l (1 = V, 2 = W; PCMGSetCycleTypeOnLevel) */
    /* coarsest level solved exactly (the reference's coarse KSP runs to rtol 1e-8, LinearElasticity.cc:628-632; the
     * product's csrc/coarse_direct.h): banded Cholesky factor, row i holds columns i - hb .. i */
    int coarse_direct;
    long chol_hb;
    double *chol;
    /* diagnostic (round 6): a separate operator for the Krylov method's own products A x0, A p (NULL: A[0]); the
     * preconditioner's fine level stays A[0] */
    csr_t *Akry;
} orc_mg_t;

/* largest eigenvalue of the symmetric tridiagonal (a[0..m-1], b[0..m-2]) by
 * Sturm-sequence bisection */
static double tridiag_lmax(int m, const double *a, const double *b) {
    double lo = a[0], hi = a[0];
    for (int i = 0; i < m; i++) {
        double rad = (i > 0 ? fabs(b[i - 1]) : 0.0) + (i < m - 1 ? fabs(b[i]) : 0.0);
        if (a[i] - rad < lo) lo = a[i] - rad;
        if (a[i] + rad > hi) hi = a[i] + rad;
    }
    for (int it = 0; it < 200; it++) {
        double mid = 0.5 * (lo + hi);
        if (mid == lo || mid == hi) break;
        /* count eigenvalues < mid */
        int cnt  = 0;
        double q = a[0] - mid;
        if (q < 0) cnt++;
        for (int i = 1; i < m; i++) {
            double den = (fabs(q) < 1e-300) ? 1e-300 : q;
            q          = a[i] - mid - b[i - 1] * b[i - 1] / den;
            if (q < 0) cnt++;
        }
        if (cnt >= m) hi = mid; /* all eigenvalues below mid */
        else lo = mid;
    }
    return 0.5 * (lo + hi);
}

/* smallest eigenvalue of the symmetric tridiagonal, Sturm bisection */
static double tridiag_lmin(int m, const double *a, const double *b) {
    double lo = a[0], hi = a[0];
    for (int i = 0; i < m; i++) {
        double rad = (i > 0 ? fabs(b[i - 1]) : 0.0) + (i < m - 1 ? fabs(b[i]) : 0.0);
        if (a[i] - rad < lo) lo = a[i] - rad;
        if (a[i] + rad > hi) hi = a[i] + rad;
    }
    for (int it = 0; it < 200; it++) {
        double mid = 0.5 * (lo + hi);
        if (mid == lo || mid == hi) break;
        int cnt  = 0;
        double q = a[0] - mid;
        if (q < 0) cnt++;
        for (int i = 1; i < m; i++) {
            double den = (fabs(q) < 1e-300) ? 1e-300 : q;
            q          = a[i] - mid - b[i - 1] * b[i - 1] / den;
            if (q < 0) cnt++;
        }
        if (cnt >= 1) hi = mid; /* at least one eigenvalue below mid */
        else lo = mid;
    }
    return 0.5 * (lo + hi);
}

/* Extreme Ritz values of D^-1/2 A D^-1/2 from `nsteps` Lanczos steps with FULL
 * reorthogonalisation (classical Gram-Schmidt applied twice against all previous
 * vectors): without it the estimates depend on rounding at the 1e-5 level after a
 * few dozen steps, which would make the Chebyshev windows -- and with them the CG
 * residual history -- irreproducible between implementations. */
static double lanczos_lmax(const csr_t *A, const double *dinv, int nsteps, double *lmin_out) {
    long n = A->nrow;
    if (nsteps > 128) nsteps = 128;
    double *dis = (double *)xmalloc(sizeof(double) * (size_t)n), *w = (double *)xmalloc(sizeof(double) * (size_t)n),
           *t = (double *)xmalloc(sizeof(double) * (size_t)n);
    double *V = (double *)xmalloc(sizeof(double) * (size_t)n * (size_t)(nsteps + 1));
    double al[128] = {0}, be[128] = {0}, h[129];
    double *v0 = V;
#pragma omp parallel for schedule(static) if (n > PFOR_MIN)
    for (long i = 0; i < n; i++) {
        dis[i] = sqrt(dinv[i]);
        v0[i]  = hash_u01((uint64_t)i, 0x5eedULL) - 0.5;
    }
    double nv = vnorm(n, v0);
#pragma omp parallel for schedule(static) if (n > PFOR_MIN)
    for (long i = 0; i < n; i++) v0[i] /= nv;
    int m = 0;
    for (int j = 0; j < nsteps; j++) {
        const double *vj = V + (size_t)j * n;
        #pragma omp parallel for schedule(static) if (n > PFOR_MIN)
        for (long i = 0; i < n; i++) t[i] = dis[i] * vj[i];
        csr_spmv(A, t, w);
        #pragma omp parallel for schedule(static) if (n > PFOR_MIN)
        for (long i = 0; i < n; i++) w[i] = dis[i] * w[i];
        double alpha = 0.0;
        for (int pass = 0; pass < 2; pass++) {
            for (int q = 0; q <= j; q++) h[q] = vdot(n, V + (size_t)q * n, w);
            for (int q = 0; q <= j; q++) {
                const double *vq = V + (size_t)q * n;
                const double hq  = h[q];
#pragma omp parallel for schedule(static) if (n > PFOR_MIN)
                for (long i = 0; i < n; i++) w[i] -= hq * vq[i];
            }
            alpha += h[j];
        }
        double beta = vnorm(n, w);
        al[m] = alpha;
        be[m] = beta;
        m++;
        if (!(beta > 1e-14 * fabs(alpha))) break;
        double *vn = V + (size_t)(j + 1) * n;
#pragma omp parallel for schedule(static) if (n > PFOR_MIN)
        for (long i = 0; i < n; i++) vn[i] = w[i] / beta;
    }
    double l = tridiag_lmax(m, al, be);
    if (lmin_out) *lmin_out = tridiag_lmin(m, al, be);
    free(dis);
    free(w);
    free(t);
    free(V);
    return l;
}

/* rigorous bound  lambda_max(D^-1 A) <= lambda_max(diag(KE)^-1 KE)  for any
 * non-negative combination A = sum_e E_e KE_e; dense symmetric power iteration
 * is enough for a (8 dof)^2 matrix: use Jacobi eigenvalue sweeps instead for
 * full accuracy. */
static double elem_lambda_bound(int ed, const double *KE) {
    double S[24 * 24];
    for (int i = 0; i < ed; i++)
        for (int j = 0; j < ed; j++)
            S[i * ed + j] = 0.5 * (KE[i * ed + j] + KE[j * ed + i]) / sqrt(KE[i * ed + i] * KE[j * ed + j]);
    /* cyclic Jacobi */
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int p = 0; p < ed; p++)
            for (int q = p + 1; q < ed; q++) off += S[p * ed + q] * S[p * ed + q];
        if (off < 1e-30) break;
        for (int p = 0; p < ed; p++)
            for (int q = p + 1; q < ed; q++) {
                double apq = S[p * ed + q];
                if (fabs(apq) < 1e-300) continue;
                double th = (S[q * ed + q] - S[p * ed + p]) / (2.0 * apq);
                double t  = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < ed; k++) {
                    double akp = S[k * ed + p], akq = S[k * ed + q];
                    S[k * ed + p] = c * akp - s * akq;
                    S[k * ed + q] = s * akp + c * akq;
                }
                for (int k = 0; k < ed; k++) {
                    double apk = S[p * ed + k], aqk = S[q * ed + k];
                    S[p * ed + k] = c * apk - s * aqk;
                    S[q * ed + k] = s * apk + c * aqk;
                }
            }
    }
    double l = S[0];
    for (int i = 1; i < ed; i++)
        if (S[i * ed + i] > l) l = S[i * ed + i];
    return l;
}
ORC_API double orc_elem_lambda_bound(int ed, const double *KE) { return elem_lambda_bound(ed, KE); }

ORC_API orc_mg_t *orc_mg_create(int nx, int ny, int nz, int dof, int nlv, int nsmooth, int ncoarse, double cheb_lo,
                                double cheb_hi) {
    orc_mg_t *s = (orc_mg_t *)xcalloc(1, sizeof(orc_mg_t));
    s->nlv = nlv;
    s->dof = dof;
    s->nsmooth = nsmooth;
    s->ncoarse = ncoarse;
    s->nlanczos = 10;
    s->nlanczos_coarse = 40;
    s->cheb_lo = cheb_lo;
    s->cheb_hi = cheb_hi;
    for (int l = 0; l < MAXLV; l++) s->cycles[l] = 1;
    for (int l = 0; l < nlv; l++) {
        s->nx[l] = ((nx - 1) >> l) + 1;
        s->ny[l] = ((ny - 1) >> l) + 1;
        s->nz[l] = ((nz - 1) >> l) + 1;
        /* TopOpt.cc:183-201: each direction divisible by 2^(nlvls-1) */
        if ((((nx - 1) >> l) << l) != nx - 1 || (((ny - 1) >> l) << l) != ny - 1 || (((nz - 1) >> l) << l) != nz - 1) {
            free(s);
            return NULL;
        }
        long n  = (long)dof * s->nx[l] * s->ny[l] * s->nz[l];
        s->b[l] = (double *)xcalloc((size_t)n, sizeof(double));
        s->x[l] = (double *)xcalloc((size_t)n, sizeof(double));
        s->r[l] = (double *)xcalloc((size_t)n, sizeof(double));
        s->d[l] = (double *)xcalloc((size_t)n, sizeof(double));
    }
    for (int l = 0; l + 1 < nlv; l++) {
        s->P[l]  = interp_csr(s->nx[l + 1], s->ny[l + 1], s->nz[l + 1], dof);
        s->PT[l] = csr_transpose(s->P[l]);
    }
    return s;
}

/* Lanczos steps of the smoothing levels' eigenvalue estimates (default 10 = PETSc's -mg_levels_esteig_ksp_max_it) */
ORC_API void orc_mg_set_nlanczos(orc_mg_t *s, int n) { s->nlanczos = n > 0 ? n : 10; }
ORC_API void orc_mg_set_fine_eig(orc_mg_t *s, int mode) { s->fine_eig = mode; }
/* coarsest level: 0 = Chebyshev run of ncoarse steps, 1 = exact solve (takes effect at the next orc_mg_assemble) */
ORC_API void orc_mg_set_coarse_direct(orc_mg_t *s, int on) { s->coarse_direct = on; }

/* A = L L^T for the banded SPD matrix of the coarsest level; returns 0 or -1 (non-positive pivot) */
static int chol_band_factor(orc_mg_t *s) {
    const csr_t *A = s->A[s->nlv - 1];
    long n = A->nrow, hb = 0;
    for (long r = 0; r < n; r++)
        for (long p = A->rp[r]; p < A->rp[r + 1]; p++)
            if (r - A->ci[p] > hb) hb = r - A->ci[p];
    free(s->chol);
    s->chol_hb = hb;
    s->chol    = (double *)xcalloc((size_t)n * (size_t)(hb + 1), sizeof(double));
    double *L  = s->chol;
#define LB(i, j) L[(size_t)(i) * (size_t)(hb + 1) + (size_t)((j) - (i) + hb)]
    for (long r = 0; r < n; r++)
        for (long p = A->rp[r]; p < A->rp[r + 1]; p++)
            if (A->ci[p] <= r) LB(r, A->ci[p]) = A->v[p];
    for (long i = 0; i < n; i++) {
        long j0 = i - hb < 0 ? 0 : i - hb;
        for (long j = j0; j <= i; j++) {
            long k0  = j - hb < j0 ? j0 : j - hb;
            double t = LB(i, j);
            for (long k = k0; k < j; k++) t -= LB(i, k) * LB(j, k);
            if (j < i) {
                LB(i, j) = t / LB(j, j);
            } else {
                if (!(t > 0.0)) return -1;
                LB(i, i) = sqrt(t);
            }
        }
    }
    return 0;
}
static void chol_band_solve(const orc_mg_t *s, const double *b, double *x) {
    long n = s->A[s->nlv - 1]->nrow, hb = s->chol_hb;
    const double *L = s->chol;
    for (long i = 0; i < n; i++) {
        long j0  = i - hb < 0 ? 0 : i - hb;
        double t = b[i];
        for (long j = j0; j < i; j++) t -= LB(i, j) * x[j];
        x[i] = t / LB(i, i);
    }
    for (long i = n - 1; i >= 0; i--) {
        long j1  = i + hb > n - 1 ? n - 1 : i + hb;
        double t = x[i];
        for (long j = i + 1; j <= j1; j++) t -= LB(j, i) * x[j];
        x[i] = t / LB(i, i);
    }
#undef LB
}
/* CPU baseline only: from now on the fine-level operator of THIS hierarchy (dof 3) is applied matrix-free from KE, E, N
 * (caller keeps the arrays alive); NULL KE switches back to the assembled matrix.  One hierarchy at a time. */
ORC_API void orc_mg_fine_matfree(orc_mg_t *s, const double *KE, const double *E, const double *N) {
    if (!KE || !s || s->dof != 3) {
        g_mf_A = NULL;
        return;
    }
    g_mf_A = s->A[0];
    g_mf_KE = KE, g_mf_E = E, g_mf_N = N;
    g_mf_nx = s->nx[0], g_mf_ny = s->ny[0], g_mf_nz = s->nz[0];
}
/* PCMGSetCycleType / PCMGSetCycleTypeOnLevel: c[l] cycles of level l + 1 per visit of level l (l = 0 finest) */
ORC_API void orc_mg_set_cycles(orc_mg_t *s, const int *c) {
    for (int l = 0; l + 1 < s->nlv; l++) s->cycles[l] = c[l] < 1 ? 1 : c[l];
}

ORC_API void orc_mg_destroy(orc_mg_t *s) {
    if (!s) return;
    for (int l = 0; l < s->nlv; l++) {
        csr_free(s->A[l]);
        csr_free(s->P[l]);
        csr_free(s->PT[l]);
        free(s->dinv[l]);
        free(s->b[l]);
        free(s->x[l]);
        free(s->r[l]);
        free(s->d[l]);
    }
    free(s->chol);
    csr_free(s->Akry);
    free(s);
}
/* Diagnostic: the operator of the KRYLOV METHOD (initial residual, A p) from another element matrix than
 * the one the preconditioner's fine level was assembled from; KE = NULL removes it.  Which of the two uses of the fine-level
 * operator carries the sensitivity of the late residual history to the element matrix's rounding residue? */
ORC_API void orc_mg_set_krylov_operator(orc_mg_t *s, const double *KE, const double *E, const double *N) {
    csr_free(s->Akry);
    s->Akry = KE ? assemble_csr(s->nx[0], s->ny[0], s->nz[0], s->dof, KE, E, N) : NULL;
}

/* Diagnostic: replace ONLY the fine-level operator and its Jacobi diagonal by the one assembled from another
 * element matrix; the coarse operators, their diagonals and every Chebyshev window stay those of the last orc_mg_assemble.
 * Separates what a change of the element matrix does through the operator of the Krylov method from what it does through
 * the preconditioner's hierarchy. */
ORC_API void orc_mg_reassemble_fine(orc_mg_t *s, const double *KE, const double *E, const double *N) {
    csr_free(s->A[0]);
    free(s->dinv[0]);
    s->A[0] = assemble_csr(s->nx[0], s->ny[0], s->nz[0], s->dof, KE, E, N);
    csr_t *A   = s->A[0];
    s->dinv[0] = (double *)xmalloc(sizeof(double) * (size_t)A->nrow);
#pragma omp parallel for schedule(static) if (A->nrow > PFOR_MIN)
    for (long r = 0; r < A->nrow; r++) {
        double dg = 0.0;
        for (long p = A->rp[r]; p < A->rp[r + 1]; p++)
            if (A->ci[p] == r) dg = A->v[p];
        s->dinv[0][r] = 1.0 / dg;
    }
}

/* "KSPSetOperators + KSPSetUp": assemble the fine matrix, Galerkin coarse
 * operators, Jacobi diagonals and Chebyshev windows (LinearElasticity.cc:190-200) */
ORC_API void orc_mg_assemble(orc_mg_t *s, const double *KE, const double *E, const double *N) {
    for (int l = 0; l < s->nlv; l++) {
        csr_free(s->A[l]);
        s->A[l] = NULL;
        free(s->dinv[l]);
        s->dinv[l] = NULL;
    }
    s->A[0] = assemble_csr(s->nx[0], s->ny[0], s->nz[0], s->dof, KE, E, N);
    for (int l = 0; l + 1 < s->nlv; l++) {
        csr_t *AP = csr_matmul(s->A[l], s->P[l]);
        s->A[l + 1] = csr_matmul(s->PT[l], AP);
        csr_free(AP);
    }
    for (int l = 0; l < s->nlv; l++) {
        csr_t *A   = s->A[l];
        s->dinv[l] = (double *)xmalloc(sizeof(double) * (size_t)A->nrow);
#pragma omp parallel for schedule(static) if (A->nrow > PFOR_MIN)
        for (long r = 0; r < A->nrow; r++) {
            double dg = 0.0;
            for (long p = A->rp[r]; p < A->rp[r + 1]; p++)
                if (A->ci[p] == r) dg = A->v[p];
            s->dinv[l][r] = 1.0 / dg;
        }
        if (l == 0 && !s->fine_eig) {
            double lb = elem_lambda_bound(8 * s->dof, KE);
            s->lam[0] = lb > 1.0 ? lb : 1.0;
        } else if (l == s->nlv - 1 && l > 0 && s->coarse_direct) {
            s->lam[l] = s->lam_min[l] = 1.0; /* not used */
            if (chol_band_factor(s)) s->lam[l] = s->lam_min[l] = NAN;
        } else if (l == s->nlv - 1 && l > 0) {
            /* coarsest level: the Chebyshev iteration there is a SOLVE (the reference uses a Krylov
             * method, LinearElasticity.cc:720-731), so its window spans the whole spectrum:
             * both extreme Ritz values of a longer Lanczos run */
            s->lam[l] = lanczos_lmax(A, s->dinv[l], s->nlanczos_coarse, &s->lam_min[l]);
        } else {
            s->lam[l] = lanczos_lmax(A, s->dinv[l], s->nlanczos, NULL);
        }
    }
}

/* Chebyshev semi-iteration, Jacobi preconditioned, k steps, each step recomputes
 * r = b - A x.  zero_guess: x is taken as 0 (first residual is b). */
static void cheb_smooth(const csr_t *A, const double *dinv, const double *b, double *x, double *r, double *d, int k,
                        double lmin, double lmax, int zero_guess) {
    long n       = A->nrow;
    double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta, rho = 1.0 / sigma;
    if (zero_guess) {
#pragma omp parallel for schedule(static) if (n > PFOR_MIN)
        for (long i = 0; i < n; i++) {
            d[i] = dinv[i] * b[i] / theta;
            x[i] = d[i];
        }
    } else {
        csr_spmv(A, x, r);
#pragma omp parallel for schedule(static) if (n > PFOR_MIN)
        for (long i = 0; i < n; i++) {
            d[i] = dinv[i] * (b[i] - r[i]) / theta;
            x[i] += d[i];
        }
    }
    for (int it = 1; it < k; it++) {
        double rn = 1.0 / (2.0 * sigma - rho);
        double c1 = rn * rho, c2 = 2.0 * rn / delta;
        csr_spmv(A, x, r);
#pragma omp parallel for schedule(static) if (n > PFOR_MIN)
        for (long i = 0; i < n; i++) {
            d[i] = c1 * d[i] + c2 * (dinv[i] * (b[i] - r[i]));
            x[i] += d[i];
        }
        rho = rn;
    }
}

/* PCMGMCycle_Private of PETSc 3.11 (multiplicative): pre-smooth, residual, restrict, the coarse iterate zeroed ONCE
 * and the coarser level cycled cycles[l] times on the same right-hand side (the second cycle starts from the first
 * one's result) -- one cycle only into the coarsest level (`cycles = (level == 1) ? 1 : mglevels->cycles`) --,
 * correction, post-smooth.  zero_guess: x[l] is taken as 0 (every level's first visit in a PCApply) */
static void mcycle(orc_mg_t *s, int l, int zero_guess) {
    const csr_t *A = s->A[l];
    long n         = A->nrow;
    double lmin = s->cheb_lo * s->lam[l], lmax = s->cheb_hi * s->lam[l];
    if (l == s->nlv - 1) {
        if (l > 0 && s->coarse_direct) {
            chol_band_solve(s, s->b[l], s->x[l]);
            return;
        }
        if (l > 0) lmin = s->lam_min[l];
        cheb_smooth(A, s->dinv[l], s->b[l], s->x[l], s->r[l], s->d[l], s->ncoarse, lmin, lmax, zero_guess);
        return;
    }
    cheb_smooth(A, s->dinv[l], s->b[l], s->x[l], s->r[l], s->d[l], s->nsmooth, lmin, lmax, zero_guess);
    csr_spmv(A, s->x[l], s->r[l]);
#pragma omp parallel for schedule(static) if (n > PFOR_MIN)
    for (long i = 0; i < n; i++) s->r[l][i] = s->b[l][i] - s->r[l][i];
    csr_spmv(s->PT[l], s->r[l], s->b[l + 1]); /* restriction = P^T (stored transpose: same sums in the same order as the scatter) */
    const int cyc = (l + 1 == s->nlv - 1) ? 1 : s->cycles[l];
    for (int c = 0; c < cyc; c++) mcycle(s, l + 1, c == 0);
    csr_spmv(s->P[l], s->x[l + 1], s->r[l]);
#pragma omp parallel for schedule(static) if (n > PFOR_MIN)
    for (long i = 0; i < n; i++) s->x[l][i] += s->r[l][i];
    cheb_smooth(A, s->dinv[l], s->b[l], s->x[l], s->r[l], s->d[l], s->nsmooth, lmin, lmax, 0);
}
static void vcycle(orc_mg_t *s, int l) { mcycle(s, l, 1); }

/* z = M r */
ORC_API void orc_mg_precond(orc_mg_t *s, const double *r, double *z) {
    long n = s->A[0]->nrow;
    memcpy(s->b[0], r, sizeof(double) * (size_t)n);
    vcycle(s, 0);
    memcpy(z, s->x[0], sizeof(double) * (size_t)n);
}

/* KSPSolve: preconditioned CG, warm start from x (LinearElasticity.cc:204, :647).
 * hist[0..its] receives ||b - A x_k||_2.  Returns the iteration count, or -1
 * on divergence (rnorm > dtol*||b||).  use_pc=0 runs Jacobi-less plain CG
 * (used as an independent cross-check in the tests). */
ORC_API int orc_mg_solve(orc_mg_t *s, const double *b, double *x, double rtol, double atol, double dtol, int maxit,
                         int use_pc, double *hist, double *rnorm_out) {
    const csr_t *A = s->Akry ? s->Akry : s->A[0];
    long n         = A->nrow;
    double *r = (double *)xmalloc(sizeof(double) * (size_t)n), *z = (double *)xmalloc(sizeof(double) * (size_t)n),
           *p = (double *)xmalloc(sizeof(double) * (size_t)n), *w = (double *)xmalloc(sizeof(double) * (size_t)n);
    csr_spmv(A, x, r);
#pragma omp parallel for schedule(static) if (n > PFOR_MIN)
    for (long i = 0; i < n; i++) r[i] = b[i] - r[i];
    double bnorm = vnorm(n, b);
    double ttol  = fmax(rtol * bnorm, atol);
    double rnorm = vnorm(n, r);
    int its      = 0;
    if (hist) hist[0] = rnorm;
    if (rnorm > ttol) {
        if (use_pc) orc_mg_precond(s, r, z);
        else memcpy(z, r, sizeof(double) * (size_t)n);
        memcpy(p, z, sizeof(double) * (size_t)n);
        double rz = vdot(n, r, z);
        for (its = 1; its <= maxit; its++) {
            csr_spmv(A, p, w);
            double alpha = rz / vdot(n, p, w);
#pragma omp parallel for schedule(static) if (n > PFOR_MIN)
            for (long i = 0; i < n; i++) {
                x[i] += alpha * p[i];
                r[i] -= alpha * w[i];
            }
            rnorm = vnorm(n, r);
            if (hist) hist[its] = rnorm;
            if (rnorm <= ttol) break;
            if (rnorm > dtol * bnorm) {
                its = -1;
                break;
            }
            if (its == maxit) break;
            if (use_pc) orc_mg_precond(s, r, z);
            else memcpy(z, r, sizeof(double) * (size_t)n);
            double rzn  = vdot(n, r, z);
            double beta = rzn / rz;
#pragma omp parallel for schedule(static) if (n > PFOR_MIN)
            for (long i = 0; i < n; i++) p[i] = z[i] + beta * p[i];
            rz = rzn;
        }
    }
    if (rnorm_out) *rnorm_out = rnorm;
    free(r);
    free(z);
    free(p);
    free(w);
    return its;
}

/* introspection for the parity tests */
ORC_API long orc_mg_level_size(orc_mg_t *s, int l) { return s->A[l] ? s->A[l]->nrow : 0; }
ORC_API long orc_mg_level_nnz(orc_mg_t *s, int l) { return s->A[l] ? s->A[l]->rp[s->A[l]->nrow] : 0; }
ORC_API double orc_mg_level_lambda(orc_mg_t *s, int l) { return s->lam[l]; }
ORC_API double orc_mg_level_lambda_min(orc_mg_t *s, int l) { return s->lam_min[l]; }
ORC_API void orc_mg_level_apply(orc_mg_t *s, int l, const double *u, double *y) { csr_spmv(s->A[l], u, y); }
ORC_API void orc_mg_level_diag(orc_mg_t *s, int l, double *d) {
    for (long i = 0; i < s->A[l]->nrow; i++) d[i] = 1.0 / s->dinv[l][i];
}
ORC_API void orc_mg_level_csr(orc_mg_t *s, int l, long *rp, int *ci, double *v) {
    const csr_t *A = s->A[l];
    memcpy(rp, A->rp, sizeof(long) * (size_t)(A->nrow + 1));
    memcpy(ci, A->ci, sizeof(int) * (size_t)A->rp[A->nrow]);
    memcpy(v, A->v, sizeof(double) * (size_t)A->rp[A->nrow]);
}
ORC_API void orc_mg_prolong(orc_mg_t *s, int l, const double *xc, double *xf) { csr_spmv(s->P[l], xc, xf); }
ORC_API void orc_mg_restrict(orc_mg_t *s, int l, const double *rf, double *rc) { csr_spmv_t(s->P[l], rf, rc); }
ORC_API void orc_mg_smooth(orc_mg_t *s, int l, const double *b, double *x, int k, int zero_guess) {
    /* same windows as the V-cycle: the coarsest level spans the whole spectrum */
    double lmin = (l == s->nlv - 1 && l > 0) ? s->lam_min[l] : s->cheb_lo * s->lam[l];
    cheb_smooth(s->A[l], s->dinv[l], b, x, s->r[l], s->d[l], k, lmin, s->cheb_hi * s->lam[l], zero_guess);
}

/* ------------------------------------------------------------------------- */
/* a7. objective, volume constraint and sensitivities                         */
/* (LinearElasticity.cc:405-437): uKu with the reference's k,h loop order.    */
/* ------------------------------------------------------------------------- */
ORC_API void orc_compliance_sens(int nx, int ny, int nz, const double *KE, const double *U, const double *xPhys,
                                 double Emin, double Emax, double penal, double volfrac, double *fx, double *gx,
                                 double *dfdx, double *dgdx) {
    int ex = nx - 1, ey = ny - 1, ez = nz - 1;
    long nel = (long)ex * ey * ez;
    double f = 0.0;
    /* element contributions in parallel, added up afterwards in the reference's element order (LinearElasticity.cc:421):
     * the same sum, bit for bit, as the serial loop */
    double *fe = (double *)xmalloc(sizeof(double) * (size_t)nel);
#pragma omp parallel for collapse(2) schedule(static) if (nel > 20000)
    for (int k = 0; k < ez; k++)
        for (int j = 0; j < ey; j++)
            for (int i = 0; i < ex; i++) {
                long nd[8];
                long e = (long)i + (long)ex * (j + (long)ey * k);
                elem_nodes(nx, ny, i, j, k, nd);
                double ue[24];
                for (int a = 0; a < 8; a++)
                    for (int c = 0; c < 3; c++) ue[3 * a + c] = U[3 * nd[a] + c];
                double uKu = 0.0;
                for (int kk = 0; kk < 24; kk++)
                    for (int hh = 0; hh < 24; hh++) uKu += ue[kk] * KE[kk * 24 + hh] * ue[hh];
                fe[e] = (Emin + pow(xPhys[e], penal) * (Emax - Emin)) * uKu;
                if (dfdx) dfdx[e] = -1.0 * penal * pow(xPhys[e], penal - 1) * (Emax - Emin) * uKu;
            }
    for (long e = 0; e < nel; e++) f += fe[e];
    free(fe);
    *fx = f;
    if (gx) *gx = vsum(nel, xPhys) / ((double)nel) - volfrac;
    if (dgdx)
        for (long e = 0; e < nel; e++) dgdx[e] = 1.0 / ((double)nel);
}

/* ------------------------------------------------------------------------- */
/* a8-a11. density / sensitivity filter with an explicit matrix H             */
/* (Filter.cc:290-463 set-up, :60-117 forward, :120-204 gradients).           */
/* ------------------------------------------------------------------------- */
typedef struct {
    int ex, ey, ez, conn;
    double R;
    csr_t *H;
    double *Hs;
} orc_filter_t;

ORC_API orc_filter_t *orc_filter_create(int nx, int ny, int nz, double dx, double dy, double dz, double R) {
    orc_filter_t *f = (orc_filter_t *)xcalloc(1, sizeof(orc_filter_t));
    int M = nx, Nn = ny, P = nz;
    f->ex = M - 1;
    f->ey = Nn - 1;
    f->ez = P - 1;
    f->R  = R;
    /* Filter.cc:326-327 */
    int conn = (int)fmax(ceil(R / dx) - 1, fmax(ceil(R / dy) - 1, ceil(R / dz) - 1));
    int cap  = (M - 1) / 2;
    if ((Nn - 1) / 2 < cap) cap = (Nn - 1) / 2;
    if ((P - 1) / 2 < cap) cap = (P - 1) / 2;
    if (conn > cap) conn = cap;
    f->conn = conn;
    int ex = f->ex, ey = f->ey, ez = f->ez;
    long nel = (long)ex * ey * ez;
    /* element-centre coordinates as DMDASetUniformCoordinates lays them out
     * between dx/2 and xmax-dx/2 (Filter.cc:375-379) */
    double x0 = dx / 2.0, x1 = (M - 1) * dx - dx / 2.0, y0 = dy / 2.0, y1 = (Nn - 1) * dy - dy / 2.0, z0 = dz / 2.0,
           z1 = (P - 1) * dz - dz / 2.0;
    double hx = ex > 1 ? (x1 - x0) / (ex - 1) : 0.0, hy = ey > 1 ? (y1 - y0) / (ey - 1) : 0.0,
           hz = ez > 1 ? (z1 - z0) / (ez - 1) : 0.0;
    csr_t *H = (csr_t *)xcalloc(1, sizeof(csr_t));
    H->nrow = H->ncol = nel;
    H->rp             = (long *)xcalloc((size_t)nel + 1, sizeof(long));
    long cap_nnz      = nel * (long)(2 * conn + 1) * (2 * conn + 1) * (2 * conn + 1);
    H->ci             = (int *)xmalloc(sizeof(int) * (size_t)cap_nnz);
    H->v              = (double *)xmalloc(sizeof(double) * (size_t)cap_nnz);
    long nnz          = 0;
    for (int k = 0; k < ez; k++)
        for (int j = 0; j < ey; j++)
            for (int i = 0; i < ex; i++) {
                long row = (long)i + (long)ex * (j + (long)ey * k);
                double cr[3] = {x0 + hx * i, y0 + hy * j, z0 + hz * k};
                for (int k2 = (k - conn > 0 ? k - conn : 0); k2 <= (k + conn < ez - 1 ? k + conn : ez - 1); k2++)
                    for (int j2 = (j - conn > 0 ? j - conn : 0); j2 <= (j + conn < ey - 1 ? j + conn : ey - 1); j2++)
                        for (int i2 = (i - conn > 0 ? i - conn : 0); i2 <= (i + conn < ex - 1 ? i + conn : ex - 1);
                             i2++) {
                            double cc[3] = {x0 + hx * i2, y0 + hy * j2, z0 + hz * k2};
                            double dist  = 0.0;
                            for (int kk = 0; kk < 3; kk++) dist = dist + pow(cr[kk] - cc[kk], 2.0);
                            dist = sqrt(dist);
                            if (dist < R) { /* strict, Filter.cc:430 */
                                H->ci[nnz] = (int)((long)i2 + (long)ex * (j2 + (long)ey * k2));
                                H->v[nnz]  = R - dist;
                                nnz++;
                            }
                        }
                H->rp[row + 1] = nnz;
            }
    f->H  = H;
    f->Hs = (double *)xmalloc(sizeof(double) * (size_t)nel);
    double *one = (double *)xmalloc(sizeof(double) * (size_t)nel);
    for (long e = 0; e < nel; e++) one[e] = 1.0;
    csr_spmv(H, one, f->Hs); /* Filter.cc:445-448 */
    free(one);
    return f;
}
ORC_API void orc_filter_destroy(orc_filter_t *f) {
    if (!f) return;
    csr_free(f->H);
    free(f->Hs);
    free(f);
}
ORC_API int orc_filter_conn(orc_filter_t *f) { return f->conn; }
ORC_API long orc_filter_nnz(orc_filter_t *f) { return f->H->rp[f->H->nrow]; }
ORC_API void orc_filter_hs(orc_filter_t *f, double *hs) { memcpy(hs, f->Hs, sizeof(double) * (size_t)f->H->nrow); }

/* Filter.h:80-88 */
static double smooth_proj(double x, double beta, double eta) {
    return (tanh(beta * eta) + tanh(beta * (x - eta))) / (tanh(beta * eta) + tanh(beta * (1.0 - eta)));
}
static double smooth_proj_d(double x, double beta, double eta) {
    return beta * (1.0 - pow(tanh(beta * (x - eta)), 2.0)) / (tanh(beta * eta) + tanh(beta * (1.0 - eta)));
}
ORC_API void orc_heaviside(long n, const double *xt, double beta, double eta, double *y) {
    for (long i = 0; i < n; i++) y[i] = smooth_proj(xt[i], beta, eta);
}
ORC_API void orc_heaviside_chain(long n, const double *xt, double beta, double eta, double *y) {
    for (long i = 0; i < n; i++) y[i] = smooth_proj_d(xt[i], beta, eta);
}
/* Filter.cc:206-225 */
ORC_API double orc_mnd(long n, const double *x) {
    double s = 0.0;
    for (long i = 0; i < n; i++) s += 4 * x[i] * (1.0 - x[i]);
    return s / (double)n;
}

/* Filter::FilterProject for filterType 0/1 (Filter.cc:60-117); type 2 is
 * orc_pdef_apply + clamp, driven from the caller. */
ORC_API void orc_filter_project(orc_filter_t *f, int type, const double *x, double *xTilde, double *xPhys, int proj,
                                double beta, double eta) {
    long n = f->H->nrow;
    if (type == 1) {
        csr_spmv(f->H, x, xTilde);
        for (long i = 0; i < n; i++) xTilde[i] = xTilde[i] / f->Hs[i];
    } else {
        memcpy(xTilde, x, sizeof(double) * (size_t)n);
    }
    if (proj) orc_heaviside(n, xTilde, beta, eta, xPhys);
    else memcpy(xPhys, xTilde, sizeof(double) * (size_t)n);
}

/* Filter::Gradients for filterType 0/1 (Filter.cc:120-192), one vector at a
 * time: call once for dfdx and once per dgdx[i]. */
ORC_API void orc_filter_gradient(orc_filter_t *f, int type, const double *x, const double *xTilde, double *df,
                                 int proj, double beta, double eta) {
    long n      = f->H->nrow;
    double *tmp = (double *)xmalloc(sizeof(double) * (size_t)n);
    if (proj)
        for (long i = 0; i < n; i++) df[i] = df[i] * smooth_proj_d(xTilde[i], beta, eta);
    if (type == 0) {
        for (long i = 0; i < n; i++) tmp[i] = df[i] * x[i];
        csr_spmv(f->H, tmp, df);
        for (long i = 0; i < n; i++) tmp[i] = df[i] / f->Hs[i];
        for (long i = 0; i < n; i++) df[i] = tmp[i] / x[i];
    } else if (type == 1) {
        for (long i = 0; i < n; i++) tmp[i] = df[i] / f->Hs[i];
        csr_spmv(f->H, tmp, df);
    }
    free(tmp);
}

/* ------------------------------------------------------------------------- */
/* a12-a14. Helmholtz PDE filter  x~ = T^T K_f^-1 (elemVol * T x)             */
/* (PDEFilter.cc:28-218).  Solver: the same CG + Galerkin-MG skeleton with    */
/* 3 levels, scalar unknowns, rtol 1e-8, <= 60 iterations (:276-285, :32).    */
/* ------------------------------------------------------------------------- */
typedef struct {
    int nx, ny, nz;
    double elemVol, R;
    double KF[64], TF[8];
    orc_mg_t *mg;
    double *rhs, *u;
    int last_its;
    double last_rnorm;
} orc_pdef_t;

ORC_API orc_pdef_t *orc_pdef_create(int nx, int ny, int nz, double dx, double dy, double dz, double rmin, int nlv,
                                    int nsmooth, int ncoarse, double cheb_lo, double cheb_hi) {
    orc_pdef_t *p = (orc_pdef_t *)xcalloc(1, sizeof(orc_pdef_t));
    p->nx = nx;
    p->ny = ny;
    p->nz = nz;
    p->R       = rmin / 2.0 / sqrt(3); /* PDEFilter.cc:30 */
    p->elemVol = dx * dy * dz;
    orc_pde_kf(dx, dy, dz, p->R, p->KF, p->TF);
    p->mg = orc_mg_create(nx, ny, nz, 1, nlv, nsmooth, ncoarse, cheb_lo, cheb_hi);
    if (!p->mg) {
        free(p);
        return NULL;
    }
    orc_mg_assemble(p->mg, p->KF, NULL, NULL);
    long nn = (long)nx * ny * nz;
    p->rhs  = (double *)xmalloc(sizeof(double) * (size_t)nn);
    p->u    = (double *)xmalloc(sizeof(double) * (size_t)nn);
    return p;
}
ORC_API void orc_pdef_destroy(orc_pdef_t *p) {
    if (!p) return;
    orc_mg_destroy(p->mg);
    free(p->rhs);
    free(p->u);
    free(p);
}
ORC_API void orc_pdef_kf(orc_pdef_t *p, double *KF) { memcpy(KF, p->KF, sizeof(p->KF)); }
ORC_API int orc_pdef_last_its(orc_pdef_t *p) { return p->last_its; }
ORC_API double orc_pdef_last_rnorm(orc_pdef_t *p) { return p->last_rnorm; }

/* PDEFilt::FilterProject (PDEFilter.cc:189-216); Gradients is the same
 * operator (:218).  in and out may alias. */
ORC_API int orc_pdef_apply(orc_pdef_t *p, const double *in, double *out, double rtol, int maxit, double *hist) {
    int nx = p->nx, ny = p->ny, nz = p->nz, ex = nx - 1, ey = ny - 1, ez = nz - 1;
    long nn = (long)nx * ny * nz;
    long nd[8];
    for (long n = 0; n < nn; n++) p->rhs[n] = 0.0;
    for (int k = 0; k < ez; k++) /* RHS = T x */
        for (int j = 0; j < ey; j++)
            for (int i = 0; i < ex; i++) {
                long e = (long)i + (long)ex * (j + (long)ey * k);
                elem_nodes(nx, ny, i, j, k, nd);
                for (int a = 0; a < 8; a++) p->rhs[nd[a]] += p->TF[a] * in[e];
            }
    for (long n = 0; n < nn; n++) {
        p->u[n]   = p->rhs[n];               /* initial guess, :200 */
        p->rhs[n] = p->rhs[n] * p->elemVol;  /* :202 */
    }
    p->last_its = orc_mg_solve(p->mg, p->rhs, p->u, rtol, 1e-50, 1e3, maxit, 1, hist, &p->last_rnorm);
    for (int k = 0; k < ez; k++) /* out = T^T U */
        for (int j = 0; j < ey; j++)
            for (int i = 0; i < ex; i++) {
                long e = (long)i + (long)ex * (j + (long)ey * k);
                elem_nodes(nx, ny, i, j, k, nd);
                double s = 0.0;
                for (int a = 0; a < 8; a++) s += p->TF[a] * p->u[nd[a]];
                out[e] = s;
            }
    return p->last_its;
}

/* bound check + clamp of Filter.cc:76-100; returns the number of violations
 * larger than 1e-4 (the reference prints a warning for each) */
ORC_API long orc_pdef_clamp(long n, double *xt) {
    long viol = 0;
    for (long i = 0; i < n; i++) {
        if (xt[i] < 0.0) {
            if (fabs(xt[i]) > 1.0e-4) viol++;
            xt[i] = 0.0;
        }
        if (xt[i] > 1.0) {
            if (fabs(xt[i] - 1.0) > 1.0e-4) viol++;
            xt[i] = 1.0;
        }
    }
    return viol;
}

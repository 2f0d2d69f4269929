// mg.h -- CG preconditioned by a geometric-multigrid V-cycle with
// Chebyshev-Jacobi smoothing, for DOF unknowns per node (3 = elasticity,
// 1 = Helmholtz filter).  Replaces KSPSolve(KSPCG) + PCApply_MG + the level
// KSPCHEBYSHEV/PCJACOBI smoothers the reference reaches through PETSc
// (LinearElasticity.cc:617-746, PDEFilter.cc:269-417).
#pragma once
#include <algorithm>
#include <thread>

#include "galerkin.h"
#include "grid.h"
#include "operators.h"
#include "coarse_run.h"
#include "matfree_tile.h"
#include "fine_tile.h"
#include "fine_u4.h"
#include "coarse_direct.h"

enum { LV_MATFREE = 0, LV_DIA = 1, LV_MACRO = 2 };

template <int DOF>
struct Level {
    Geom g;
    int kind;
    // matrix-free levels
    const double *KE;       // [dev] (8 DOF)^2
    const double *E;        // [dev] per stored element, or null
    const uint8_t *mask;    // [dev] per node, or null
    // stencil levels
    double *S;              // [dev] 27*DOF diagonals x (DOF*nodes)
    double *Kel;            // [dev] coarse element matrices (DOF = 3 Galerkin levels)
    double *dinv;
    double lam;             // estimate / bound of lambda_max(D^-1 A)
    double lam_min = 0.0;   // coarsest level: smallest Ritz value (coarse-solve window)
    double *b, *x, *x2, *r, *d;
    bool no_comm = false;   // replicated (global) copy of the coarsest level: no halo, no reductions over ranks
    bool use_tile = false;  // DOF == 3 matrix-free level with a box-symmetric KE: tuned kernel
    int sym_slot = -1;                // slot of the packed SymKE in constant memory
    // LV_MACRO (level 1 applied from the fine densities)
    int fex = 0, fey = 0;             // fine element counts
    // Dirichlet correction (elements containing a clamped fine node)
    const double *dK = nullptr;       // [dev] nflag x 576
    const int *flag_list = nullptr;   // [dev] flagged stored coarse elements
    const int *corr_nodes = nullptr;  // [dev] affected owned nodes
    const int *corr_adj = nullptr;    // [dev] 8 per affected node
    int ncorr_nodes = 0, nflag = 0;
    double *corr_tmp = nullptr;       // [dev] nflag x 24 element-row products
    double *corr = nullptr;           // [dev] level dofs, zero outside the affected nodes
    const uint8_t *colmask = nullptr; // [dev] per node column: OR of mask over z
    const double *wtab = nullptr;     // [dev] DOF == 1, constant coefficients: 27 x 27 stencil table (operators.h: ScalarStencilOp)
    long ndof() const { return (long)DOF * g.nodes(); }
    long own_off() const { return (long)DOF * g.plane() * g.own_lo; }
    long own_n() const { return (long)DOF * g.owned_nodes(); }
};

// CG scalar slots in tp_grid::scal
enum { S_BB = 0, S_RR = 1, S_PW = 2, S_RZ0 = 3, S_RZ1 = 4, S_TMP = 8 };

// x += alpha p, r -= alpha w, ||r||^2 (also to pinned host memory if out_host); z1 != NULL: also the first Chebyshev step of
// the NEXT V-cycle's pre-smoothing from its zero guess, z1 = dinv r / theta (k_cheb_first: the same product in the same
// order) -- one pass over r less and one launch less per Krylov iteration
// NT: x, p and w pass through with non-temporal loads / stores -- none of the three is read again before ~10 other vectors
// of the same size have gone by, while r and z1 are the next kernel's input: the hint keeps the streamed ones from
// displacing them in the Infinity Cache.  Measured at 128^3 (two runs each, round 5): 12.50 / 12.48 -> 12.40 / 12.32 ms per
// design iteration, the following fine-level Chebyshev launches 58.9 / 57.3 -> 54.7 / 54.4 us (in-step roofline fraction
// 0.40-0.41 -> 0.43).  Measured and dropped in the same round: the stores of r and z1 non-temporal as well (12.39 / 12.28 against
// 12.37 / 12.23: the next kernel then misses them), the restriction reading the fine residual non-temporally (12.6-12.8: slower)
template <bool NT>
__global__ __launch_bounds__(BLK) void k_cg_update_xr(double *__restrict__ x, double *__restrict__ r,
                                                      const double *__restrict__ p, const double *__restrict__ w,
                                                      const double *__restrict__ scal, int slot_rz, long off, long n,
                                                      double *__restrict__ partials, unsigned *ticket,
                                                      double *__restrict__ out, double *__restrict__ out_host,
                                                      double *__restrict__ z1, const double *__restrict__ dinv, double inv_theta) {
    const double alpha = scal[slot_rz] / scal[S_PW];
    double s = 0.0;
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) {
        const long q = off + i;
        double rn;
        if constexpr (NT) {
            __builtin_nontemporal_store(fma(alpha, __builtin_nontemporal_load(p + q), __builtin_nontemporal_load(x + q)), x + q);
            rn = fma(-alpha, __builtin_nontemporal_load(w + q), r[q]);
        } else {
            x[q] = fma(alpha, p[q], x[q]);
            rn = fma(-alpha, w[q], r[q]);
        }
        r[q] = rn;
        s = fma(rn, rn, s);
        if (z1) z1[q] = dinv[q] * rn * inv_theta;
    }
    const double v[1] = {block_sum(s)};
    reduce_tail<1>(v, partials, gridDim.x, blockIdx.x, ticket, out, out_host);
}
// p = z + (rz_new/rz_old) p   (first: p = z)
__global__ __launch_bounds__(BLK) void k_cg_update_p(double *__restrict__ p, const double *__restrict__ z,
                                                     const double *__restrict__ scal, int slot_new, int slot_old,
                                                     int first, long off, long n) {
    const double beta = first ? 0.0 : scal[slot_new] / scal[slot_old];
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) {
        const long q = off + i;
        p[q] = first ? z[q] : fma(beta, p[q], z[q]);
    }
}
// two dot products in one pass: partials[b] = a1.b1, partials[nb + b] = a2.b2
__global__ __launch_bounds__(BLK) void k_dot2(const double *__restrict__ a1, const double *__restrict__ b1,
                                              const double *__restrict__ a2, const double *__restrict__ b2, long off,
                                              long n, double *__restrict__ partials) {
    double s1 = 0.0, s2 = 0.0;
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) {
        s1 = fma(a1[off + i], b1[off + i], s1);
        s2 = fma(a2[off + i], b2[off + i], s2);
    }
    s1 = block_sum(s1);
    s2 = block_sum(s2);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = s1;
        partials[gridDim.x + blockIdx.x] = s2;
    }
}

// Lanczos helpers (owned range)
template <int DOF>
__global__ __launch_bounds__(BLK) void k_lanczos_init(Geom g, double *__restrict__ v, double *__restrict__ dis,
                                                      const double *__restrict__ dinv, double *__restrict__ coef, int ncoef) {
    // the run's coefficient table starts from zero (a memset node in the replayed chain cost ~50 us before its first kernel)
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < ncoef; i += BLK) coef[i] = 0.0;
    const long plane = g.plane();
    const long t = blockIdx.x * (long)BLK + threadIdx.x;
    if (t >= g.owned_nodes()) return;
    const long n = t + plane * g.own_lo;
    const uint64_t gn = (uint64_t)(n + plane * (long)g.gz0);  // global node id
#pragma unroll
    for (int c = 0; c < DOF; c++) {
        v[n * DOF + c] = hash_u01(gn * DOF + c, 0x5eedULL) - 0.5;
        dis[n * DOF + c] = sqrt(dinv[n * DOF + c]);
    }
}
// partials[q*nb + b] = sum over block b of V_q . w   (grid = (nb, nv); V_q = V + q*stride)
// Round 6: with `mticket` the last workgroup of vector q to arrive adds q's partial sums itself, in k_reduce_multi's order
// (bitwise the same value, one dependent launch less per Gram-Schmidt pass).  Counters as in reduce_tail (common.h), one set
// of 8 shards + top per vector, MT_STRIDE words apart (a cache line of their own each); they rest at 0.
constexpr int MT_STRIDE = 64;             // unsigned words between two counters (256 B)
constexpr int MT_WORDS = 9 * MT_STRIDE;   // per vector
__global__ __launch_bounds__(BLK) void k_multi_dot(const double *__restrict__ V, long stride, int nv,
                                                   const double *__restrict__ w, long off, long n,
                                                   double *__restrict__ partials, unsigned *mticket,
                                                   double *__restrict__ out) {
    const int q = blockIdx.y;
    const double *__restrict__ vq = V + (long)q * stride;
    // 4 independent chains: with one workgroup per vector (small levels) the loop is latency bound
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const long st = (long)gridDim.x * BLK;
    long i = blockIdx.x * (long)BLK + threadIdx.x;
    for (; i + 3 * st < n; i += 4 * st) {
        s0 = fma(vq[off + i], w[off + i], s0);
        s1 = fma(vq[off + i + st], w[off + i + st], s1);
        s2 = fma(vq[off + i + 2 * st], w[off + i + 2 * st], s2);
        s3 = fma(vq[off + i + 3 * st], w[off + i + 3 * st], s3);
    }
    for (; i < n; i += st) s0 = fma(vq[off + i], w[off + i], s0);
    double s = block_sum((s0 + s1) + (s2 + s3));
    const int nb = gridDim.x, b = blockIdx.x;
    if (!mticket) {
        if (threadIdx.x == 0) partials[(long)q * nb + b] = s;
        return;
    }
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        __hip_atomic_store(&partials[(long)q * nb + b], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int sh = b & 7;
        const unsigned in_shard = (unsigned)((nb + 7 - sh) >> 3);
        unsigned *base = mticket + (size_t)q * MT_WORDS, *mine_t = base + sh * MT_STRIDE, *top_t = base + 8 * MT_STRIDE;
        int last = 0;
        if (__hip_atomic_fetch_add(mine_t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_shard - 1) {
            __hip_atomic_store(mine_t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned shards = (unsigned)(nb < 8 ? nb : 8);
            last = __hip_atomic_fetch_add(top_t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == shards - 1;
            if (last) __hip_atomic_store(top_t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    double t = 0.0;
    for (int bb = threadIdx.x; bb < nb; bb += BLK)
        t += __hip_atomic_load(&partials[(long)q * nb + bb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = block_sum(t);
    if (threadIdx.x == 0) out[q] = t;
}
// out[q] = sum_b partials[q*nb + b]; one workgroup per value (grid = nv)
__global__ __launch_bounds__(BLK) void k_reduce_multi(const double *__restrict__ partials, int nb, int nv,
                                                      double *__restrict__ out) {
    const int q = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nb; b += BLK) s += partials[(long)q * nb + b];
    s = block_sum(s);
    if (threadIdx.x == 0) out[q] = s;
}
// w -= sum_q h[q] V_q ; acc[q] += h[q]  (device-resident coefficients)
// NORM (round 6): also |w|^2 of the updated vector (the beta of the Lanczos step) -> nrm_out, finished by the last workgroup
// (reduce_tail) or, without a ticket, left as gridDim.x partial sums for k_reduce_multi.
template <bool NORM>
__global__ __launch_bounds__(BLK) void k_multi_axpy(const double *__restrict__ V, long stride, int nv,
                                                    const double *__restrict__ h, double *__restrict__ w, long off,
                                                    long n, const double *__restrict__ hprev, double *__restrict__ alpha,
                                                    double *__restrict__ nrm_part, unsigned *ticket, double *__restrict__ nrm_out) {
    // Lanczos, second Gram-Schmidt pass: alpha[j] = h1[j] + h2[j] with j = nv - 1
    if (hprev && blockIdx.x == 0 && threadIdx.x == 0) alpha[nv - 1] = hprev[nv - 1] + h[nv - 1];
    double nrm = 0.0;
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) {
        double acc = w[off + i];
        for (int q = 0; q < nv; q++) acc = fma(-h[q], V[(long)q * stride + off + i], acc);
        w[off + i] = acc;
        if (NORM) nrm = fma(acc, acc, nrm);
    }
    if (NORM) {
        const double v[1] = {block_sum(nrm)};
        reduce_tail<1>(v, nrm_part, gridDim.x, blockIdx.x, ticket, nrm_out);
    }
}
// alpha[j] = h1[j] + h2[j]
__global__ void k_lanczos_alpha(const double *__restrict__ h1, const double *__restrict__ h2, int j,
                                double *__restrict__ alpha) {
    if (threadIdx.x == 0 && blockIdx.x == 0) alpha[j] = h1[j] + h2[j];
}
// beta[j] = sqrt(bb[0]);  v_next = w / beta;  t = dis .* v_next (the scaled input of the next operator application)
__global__ __launch_bounds__(BLK) void k_lanczos_next(const double *__restrict__ w, const double *__restrict__ bb, int j,
                                                      double *__restrict__ beta, double *__restrict__ vnext, long off,
                                                      long n, const double *__restrict__ dis, double *__restrict__ t) {
    const double bt = sqrt(bb[0]);
    if (blockIdx.x == 0 && threadIdx.x == 0) beta[j] = bt;
    const double inv = bt > 0.0 ? 1.0 / bt : 0.0;
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) {
        const double v = w[off + i] * inv;
        vnext[off + i] = v;
        t[off + i] = dis[off + i] * v;
    }
}

// largest eigenvalue of a symmetric tridiagonal matrix, Sturm bisection
inline double tridiag_lmax(int m, const double *a, const double *b) {
    double lo = a[0], hi = a[0];
    for (int i = 0; i < m; i++) {
        double rad = (i > 0 ? fabs(b[i - 1]) : 0.0) + (i < m - 1 ? fabs(b[i]) : 0.0);
        lo = fmin(lo, a[i] - rad);
        hi = fmax(hi, a[i] + rad);
    }
    for (int it = 0; it < 200; it++) {
        double mid = 0.5 * (lo + hi);
        if (mid == lo || mid == hi) break;
        int cnt = 0;
        double q = a[0] - mid;
        if (q < 0) cnt++;
        for (int i = 1; i < m; i++) {
            double den = (fabs(q) < 1e-300) ? 1e-300 : q;
            q = a[i] - mid - b[i - 1] * b[i - 1] / den;
            if (q < 0) cnt++;
        }
        if (cnt >= m) hi = mid;
        else lo = mid;
    }
    return 0.5 * (lo + hi);
}

// smallest eigenvalue of a symmetric tridiagonal matrix, Sturm bisection
inline double tridiag_lmin(int m, const double *a, const double *b) {
    double lo = a[0], hi = a[0];
    for (int i = 0; i < m; i++) {
        double rad = (i > 0 ? fabs(b[i - 1]) : 0.0) + (i < m - 1 ? fabs(b[i]) : 0.0);
        lo = fmin(lo, a[i] - rad);
        hi = fmax(hi, a[i] + rad);
    }
    for (int it = 0; it < 200; it++) {
        double mid = 0.5 * (lo + hi);
        if (mid == lo || mid == hi) break;
        int cnt = 0;
        double q = a[0] - mid;
        if (q < 0) cnt++;
        for (int i = 1; i < m; i++) {
            double den = (fabs(q) < 1e-300) ? 1e-300 : q;
            q = a[i] - mid - b[i - 1] * b[i - 1] / den;
            if (q < 0) cnt++;
        }
        if (cnt >= 1) hi = mid;
        else lo = mid;
    }
    return 0.5 * (lo + hi);
}

// lambda_max(diag(KE)^-1 KE) by cyclic Jacobi rotations (n <= 24): the rigorous,
// density-independent Chebyshev bound of the matrix-free level
inline double elem_lambda_bound(int n, const double *KE) {
    std::vector<double> S((size_t)n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) S[i * n + j] = 0.5 * (KE[i * n + j] + KE[j * n + i]) / sqrt(KE[i * n + i] * KE[j * n + j]);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) off += S[p * n + q] * S[p * n + q];
        if (off < 1e-30) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = S[p * n + q];
                if (fabs(apq) < 1e-300) continue;
                double th = (S[q * n + q] - S[p * n + p]) / (2.0 * apq);
                double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = S[k * n + p], akq = S[k * n + q];
                    S[k * n + p] = c * akp - s * akq;
                    S[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = S[p * n + k], aqk = S[q * n + k];
                    S[p * n + k] = c * apk - s * aqk;
                    S[q * n + k] = s * apk + c * aqk;
                }
            }
    }
    double l = S[0];
    for (int i = 1; i < n; i++) l = fmax(l, S[i * n + i]);
    return l;
}

// Generation of the fine-level operator kernel: 3 fine_u4.h, 2 fine_tile.h, 1 matfree_tile.h; 0 (default) = by mesh
// size: the third generation from 160 tiles of 32 x 8 per z-chunk on (256x128x128 and larger), the second below --
// measured through the library on the BASELINE meshes (tools/ab3.sh): 256^3 319 / 411 us against 334 / 471 (product /
// Chebyshev step), 256x128x128 79 / 127 against 82 / 130, 128^3 41 / 65 against 41 / 60, 128x64x64 17.8 / 22.5 against
// 16.7 / 21.7.  On the small meshes one round of workgroups covers the mesh and the launch lasts as long as its slowest
// workgroup -- a tile with a Dirichlet condition, which the third generation serves ~25 % slower than a free one.
inline int fine_version() {
    static const int v = getenv("TP_FINE_V") ? atoi(getenv("TP_FINE_V")) : 0;
    return v;
}
inline int xcd_remap() {
    // on by default: tiles that share cache lines meet in one XCD's L2 (PMC: -30 % fetch traffic, -5 % time)
    static const int v = getenv("TP_XCD_REMAP") ? atoi(getenv("TP_XCD_REMAP")) : 1;
    return v;
}

// z-chunk length of the fine tile kernels.  The chip holds 768 workgroups of them at once (3 per CU: 168 VGPRs,
// 49 KB LDS).  A grid that fills those slots ONCE with equal chunks has no second, partly empty round and the least
// redundant z-halo (measured, DESIGN.md 4.1: 128^3 kz 15 -> 729 workgroups beats kz 8 -> 1377 by 10 %; 128x64x64
// kz 4 -> 765 beats kz 8 by 20 %); that only pays while the chunks stay short enough to fill >= 90 % of the slots,
// otherwise several rounds of kz ~ 8..32 are better (256x128x128: kz 8 beats kz 33).
inline int fine_kz(int planes, int tiles, int fine_v, int SLOTS = 768) {
    if (fine_v >= 2 || fine_v == 0) {
        const int tz1 = SLOTS / tiles;
        if (tz1 >= 1) {
            const int kz1 = (planes + tz1 - 1) / tz1;
            const int n1 = tiles * ((planes + kz1 - 1) / kz1);
            if (kz1 >= 3 && kz1 <= 20 && 10 * n1 >= 9 * SLOTS) return kz1;
        }
    }
    int kz = (int)((long)planes * tiles / 5120);  // ~5k workgroups
    kz = kz < 8 ? 8 : (kz > 32 ? 32 : kz);
    // small grids: shorter chunks until the workgroup slots of the chip are filled once
    if ((long)tiles * ((planes + 7) / 8) < SLOTS) kz = (int)((long)planes * tiles / SLOTS);
    const int kmin = (long)tiles * ((planes + 3) / 4) < 256 ? 2 : 4;  // 64x32x32 elements: kz 2 beats 4 by 5 % of the step
    if (kz < kmin) kz = kmin;
    return kz > planes ? planes : kz;
}

template <int DOF>
struct MGSolver;
// the reference's hard-coded FGMRES / GMRES / SOR configuration (refksp.h), tp_solver_opts::ksp_mode = 1
template <int DOF>
int refksp_solve(MGSolver<DOF> &mg, const double *b, double *x, int *its, double *rnorm, double *bnorm, double *hist, int hist_cap);
template <int DOF>
int refksp_precond(MGSolver<DOF> &mg, const double *r, double **z);
template <int DOF>
void refksp_free(MGSolver<DOF> &mg);

template <int DOF>
struct MGSolver {
    tp_grid *grid = nullptr;
    void *refksp = nullptr;  // RefKsp<DOF>: work space of the ksp_mode 1 solver
    int nlv = 0;
    // lv[0 .. nlv-1]: the levels of this rank's slab.  Several ranks: the coarse levels rep0 .. nlv-1 also exist as REPLICATED
    // global copies at lv[nlv + (l - rep0)] (no halo, no reductions over ranks: every rank runs them redundantly from one
    // all-gather of the right-hand side per visit).  rep0 = nlv - 1 (the coarsest level only) unless the slabs are thin or
    // TP_REPLICATE_FROM says otherwise (round 5: the agglomeration of the coarse levels, taken to its end).
    static constexpr int LV_SLOTS = 2 * (TP_MAX_LEVELS + 1);
    Level<DOF> lv[LV_SLOTS];
    bool replicate = false;
    int rep0 = -1;                       // first replicated level (valid if replicate)
    int rix(int l) const { return nlv + (l - rep0); }                  // slot of the replicated copy of level l >= rep0
    int base(int i) const { return i < nlv ? i : rep0 + (i - nlv); }   // level number of slot i
    bool coarsest(int i) const { return base(i) == nlv - 1; }
    bool is_rep(int i) const { return i >= nlv; }
    bool allow_replicate = false;  // set by the owner when the coarsest level is a stored stencil (elasticity)
    tp_solver_opts opt;
    double *cg_r = nullptr, *cg_p = nullptr, *cg_w = nullptr, *cg_p2 = nullptr;
    bool ready = false;
    int last_nblocks = 0;  // workgroups (= reduction partials) of the last op<EPI_APPLY_DOT>
    static constexpr int NLANCZOS_COARSE = 40;

    // Chebyshev windows of the stencil / coarse levels (and of the fine level if opt.fine_eig)
    int estimate_spectra(int first_level) {
        // One rank: the estimates of the levels are independent chains of small kernels -> one stream per level,
        // forked from and joined to the solver's stream (the device overlaps their launch-latency-bound steps).
        static const bool serial = getenv("TP_LANCZOS_SERIAL") != nullptr;
        // the coarsest level solved exactly needs no window: its factorisation takes the place of its Lanczos run
        const bool direct = coarse_direct_ok();
        const bool early = cd_early && direct;  // factorisation enqueued by the owner already (its event is recorded)
        cd_early = false;
        struct ReadyReset {  // the level events belong to this assembly only
            bool *f;
            ~ReadyReset() {
                for (int i = 0; i < LV_SLOTS; i++) f[i] = false;
            }
        } ready_reset{lv_ready_set};
        if (!early) cd.factored = false, cd_inverse_owed = false;
        if (!grid->has_comm && !serial && nlv - first_level >= 2) {
            hipStream_t main = grid->stream;
            if (!lan_fork) TP_HIP(hipEventCreateWithFlags(&lan_fork, hipEventDisableTiming));
            TP_HIP(hipEventRecord(lan_fork, main));
            int rc = TP_OK;
            static const bool serial_host = getenv("TP_LANCZOS_ONE_THREAD") != nullptr;
            struct Replay {
                hipStream_t s;
                int l;
            };
            bool chain_on_main = false;
            std::vector<Replay> replay;
            // coarsest level first: with the exact coarse solve its chain (factorisation) is the longest one
            // (with the factorisation the other levels' chains share ONE stream: the device has four hardware queues, and
            // streams that share a queue run one after the other -- the factorisation must not be the one that waits)
            for (int l = nlv - 1; l >= first_level && rc == TP_OK; l--) {
                if (early && l == nlv - 1) continue;
                // stream of this level's chain: with the factorisation, level `first_level` on one stream, the levels
                // between it and the coarsest one on the owner's spare stream (or on the same one if there is none)
                hipStream_t ls;
                if (direct && l != nlv - 1) {
                    // round 6: the levels beyond first_level + 1 run on the solver's own stream -- it has nothing else to do
                    // until the chains are in (its queue was the idle fourth one), and two levels' chains one after the other on
                    // the spare stream had become the longest path of the set-up once the chains lost their reduction launches
                    static const int on_main = getenv("TP_LANCZOS_ON_MAIN") ? atoi(getenv("TP_LANCZOS_ON_MAIN")) : 1;
                    if (l > first_level + 1 && side_stream && on_main == 1) {
                        ls = main;
                        chain_on_main = true;
                    } else if (l > first_level + 1 && side_stream && on_main == 2) {
                        if (!lan_stream[l]) TP_HIP(hipStreamCreateWithFlags(&lan_stream[l], hipStreamNonBlocking));
                        ls = lan_stream[l];
                    } else if (l != first_level && side_stream) {
                        ls = side_stream;
                    } else {
                        if (!lan_stream[first_level]) TP_HIP(hipStreamCreateWithFlags(&lan_stream[first_level], hipStreamNonBlocking));
                        ls = lan_stream[first_level];
                    }
                } else {
                    if (!lan_stream[l]) TP_HIP(hipStreamCreateWithFlags(&lan_stream[l], hipStreamNonBlocking));
                    ls = lan_stream[l];
                }
                if (!lan_done[l]) TP_HIP(hipEventCreateWithFlags(&lan_done[l], hipEventDisableTiming));
                TP_HIP(hipStreamWaitEvent(ls, (lv_ready_set[l] && l != nlv - 1) ? lv_ready[l] : lan_fork, 0));
                const int steps = (l == nlv - 1 && l > 0) ? NLANCZOS_COARSE : opt.nlanczos;
                // A captured chain is replayed from a helper thread (round 4): hipGraphLaunch of a ~100-node chain keeps
                // the calling thread for 0.6-1.3 ms (rocprofv3 --hip-trace), so three replays issued one after the other
                // made the LAST level's chain start 1-2 ms late whatever stream it was on -- the set-up was bound by
                // the host.  Chains that share a stream share a thread (order on the stream = order of the calls).
                if (!(direct && l == nlv - 1) && !serial_host && lanczos_graph_replayable(l)) {
                    lan[l].m = steps;
                    replay.push_back({ls, l});
                    continue;
                }
                grid->stream = ls;  // everything the run launches goes to the level's stream
                rc = (direct && l == nlv - 1) ? coarse_direct_factor() : lanczos_graph(l, steps);
                grid->stream = main;
                if (rc == TP_OK && hipEventRecord(lan_done[l], ls) != hipSuccess) rc = TP_ERR_HIP;
            }
            if (!replay.empty()) {
                int dev = 0;
                (void)hipGetDevice(&dev);
                std::vector<hipStream_t> streams;
                for (const Replay &r : replay)
                    if (std::find(streams.begin(), streams.end(), r.s) == streams.end()) streams.push_back(r.s);
                std::vector<int> trc(streams.size(), TP_OK);
                std::vector<std::thread> th;
                for (size_t q = 0; q < streams.size(); q++)
                    th.emplace_back([&, q, dev]() {
                        if (hipSetDevice(dev) != hipSuccess) {
                            trc[q] = TP_ERR_HIP;
                            return;
                        }
                        for (const Replay &r : replay) {
                            if (r.s != streams[q]) continue;
                            if (hipGraphLaunch(lan_graph[r.l], r.s) != hipSuccess || hipEventRecord(lan_done[r.l], r.s) != hipSuccess) trc[q] = TP_ERR_HIP;
                        }
                    });
                if (early) rc = enqueue_owed_inverse();  // (while the helper threads sit in hipGraphLaunch)
                for (std::thread &t : th) t.join();
                for (size_t q = 0; q < streams.size(); q++)
                    if (trc[q] != TP_OK) {   // a failed replay: drop the graphs, enqueue the chains the plain way
                        (void)hipGetLastError();
                        for (const Replay &r : replay) {
                            if (r.s != streams[q]) continue;
                            lan_graph_state[r.l] = -1;
                            grid->stream = r.s;
                            const int rc2 = lanczos_enqueue(r.l, lan[r.l].m);
                            grid->stream = main;
                            if (rc2 == TP_OK && hipEventRecord(lan_done[r.l], r.s) != hipSuccess) rc = TP_ERR_HIP;
                            if (rc2) rc = rc2;
                        }
                    }
            }
            if (early && rc == TP_OK) rc = enqueue_owed_inverse();  // (no replay this time: behind the directly enqueued chains)
            // Round 5: the factorisation is NOT joined here.  Nothing on the host depends on it (no Ritz values to read), and
            // the solve does not touch the factor before the first V-cycle reaches the coarsest level -- ~0.35 ms of
            // fine-level and level-1..3 work into the solve.  The solver's stream waits for the chain's event right before the
            // first triangular product (coarse_direct_apply); until then the factorisation (ONE workgroup column on one
            // XCD, 1.4 ms) runs beside the head of the solve.  TP_NO_DEFER_FACTOR=1: joined here, as in round 4.
            static const bool no_defer = getenv("TP_NO_DEFER_FACTOR") != nullptr;
            const bool defer = direct && !no_defer && !tp_defer_disabled() && lan_done[nlv - 1] && rc == TP_OK;
            for (int l = first_level; l < nlv; l++)
                if (lan_done[l] && !(defer && l == nlv - 1)) (void)hipStreamWaitEvent(main, lan_done[l], 0);
            for (int l = first_level; l < nlv; l++)
                if (lan_stream[l] && !(defer && l == nlv - 1)) (void)hipStreamSynchronize(lan_stream[l]);
            if (side_stream) (void)hipStreamSynchronize(side_stream);
            if (chain_on_main) (void)hipStreamSynchronize(main);
            cd_pending = cd_deferred_last = defer;
            if (rc) return rc;
            for (int l = first_level; l < nlv; l++) {
                if (direct && l == nlv - 1)
                    lv[l].lam = lv[l].lam_min = 1.0;  // (not used)
                else if (l == nlv - 1 && l > 0)
                    lanczos_finish(l, &lv[l].lam, &lv[l].lam_min);
                else
                    lanczos_finish(l, &lv[l].lam);
            }
            return TP_OK;
        }
        if (early) TP_TRY(enqueue_owed_inverse());
        for (int l = first_level; l < nlv; l++) {
            const bool rep = replicate && l >= rep0;   // the level's estimate comes from its replicated copy: same operator, same
            const int r = rep ? rix(l) : l;             // hashed start vector, no communication
            if (l == nlv - 1 && l > 0 && direct) {
                TP_TRY(coarse_direct_factor());
                lv[l].lam = lv[l].lam_min = 1.0;
                if (rep) lv[r].lam = lv[r].lam_min = 1.0;
            } else if (l == nlv - 1 && l > 0) {
                TP_TRY(lanczos(r, NLANCZOS_COARSE, &lv[r].lam, &lv[r].lam_min));
                lv[l].lam = lv[r].lam;
                lv[l].lam_min = lv[r].lam_min;
            } else {
                TP_TRY(lanczos(r, opt.nlanczos, &lv[r].lam));
                lv[l].lam = lv[r].lam;
            }
        }
        return TP_OK;
    }

    int alloc_levels() {
        for (int l = 0; l < nlv; l++) {
            Level<DOF> &L = lv[l];
            L.g = make_geom(grid, l);
            size_t nb = sizeof(double) * (size_t)L.ndof();
            for (double **p : {&L.b, &L.x, &L.x2, &L.r, &L.d, &L.dinv}) {
                TP_HIP(hipMalloc((void **)p, nb));
                TP_HIP(hipMemsetAsync(*p, 0, nb, grid->stream));
            }
        }
        for (int i = nlv; i < LV_SLOTS; i++) {
            lv[i] = Level<DOF>();
            lv[i].b = lv[i].x = lv[i].x2 = lv[i].r = lv[i].d = lv[i].dinv = lv[i].S = lv[i].Kel = nullptr;
        }
        static const bool no_rep = getenv("TP_NO_REPLICATE") != nullptr;
        replicate = allow_replicate && grid->has_comm && nlv > 1 && !no_rep;
        rep0 = nlv - 1;
        if (replicate) {
            // Which coarse levels are replicated?  Always the coarsest one; from the first stored-stencil level (>= 2) on whose
            // slab holds at most two element layers -- there a level's kernels are at their launch floor and every operator
            // application is followed by a halo exchange that costs more than the kernel --; TP_REPLICATE_FROM=l (2 .. nlv - 1)
            // fixes it, TP_REPLICATE_FROM=0 keeps the coarsest level only.
            static const int from_env = (getenv("TP_REPLICATE_FROM") && *getenv("TP_REPLICATE_FROM")) ? atoi(getenv("TP_REPLICATE_FROM")) : -1;
            if (from_env >= 2 && from_env <= nlv - 1) {
                // a forced level must fit: one padded slab of it in the communicator's staging buffer (gather_owned), and its
                // replicated stencil (27 DOF^2 doubles per row, on EVERY rank) within 2 GiB -- else the next thinner level
                rep0 = from_env;
                while (rep0 < nlv - 1) {
                    const Geom &c = lv[rep0].g;
                    const long pad = (long)DOF * c.plane() * (c.ez_own + 1);
                    const double rep_bytes = 8.0 * 27 * DOF * DOF * (double)c.plane() * c.nz_glob;
                    if (pad <= grid->comm.cap && rep_bytes <= 2147483648.0) break;
                    fprintf(stderr, "topopt_amd: TP_REPLICATE_FROM=%d: level %d does not fit (one slab of it: %ld doubles against a staging "
                                    "buffer of %ld; replicated stencil %.2f GB per rank, limit 2): replicating from level %d\n",
                            from_env, rep0, pad, (long)grid->comm.cap, rep_bytes / 1e9, rep0 + 1);
                    rep0++;
                }
            } else if (from_env < 0) {
                for (int l = 2; l < nlv - 1; l++)
                    if (lv[l].g.ez_own <= 2) {
                        rep0 = l;
                        break;
                    }
            }
            for (int l = rep0; l < nlv; l++) {
                Level<DOF> &R = lv[rix(l)];
                const Geom &c = lv[l].g;
                R.g = c;
                R.g.nzl = c.nz_glob;
                R.g.ez_own = R.g.ezl = c.nz_glob - 1;
                R.g.own_lo = 0;
                R.g.own_hi = c.nz_glob - 1;
                R.g.gz0 = 0;
                R.g.has_lo = R.g.has_hi = 0;
                R.kind = LV_DIA;
                R.no_comm = true;
                size_t rb = sizeof(double) * (size_t)R.ndof();
                for (double **p : {&R.b, &R.x, &R.x2, &R.r, &R.d, &R.dinv}) {
                    TP_HIP(hipMalloc((void **)p, rb));
                    TP_HIP(hipMemsetAsync(*p, 0, rb, grid->stream));
                }
                TP_HIP(hipMalloc((void **)&R.S, rb * 28 * DOF));   // (+ DOF slices: row-sum correction of the mirrored reads)
                TP_HIP(hipMemsetAsync(R.S, 0, rb * 28 * DOF, grid->stream));
            }
        }
        size_t nb = sizeof(double) * (size_t)lv[0].ndof();
        for (double **p : {&cg_r, &cg_p, &cg_w, &cg_p2}) {
            TP_HIP(hipMalloc((void **)p, nb));
            TP_HIP(hipMemsetAsync(*p, 0, nb, grid->stream));
        }
        return TP_OK;
    }
    void free_levels() {
        smooth_graphs_free();
        refksp_free(*this);
        for (int l = 0; l < LV_SLOTS; l++) {
            Level<DOF> &L = lv[l];
            if (l >= nlv && !(replicate && l < nlv + (nlv - rep0))) continue;   // unused slots
            for (double *p : {L.b, L.x, L.x2, L.r, L.d, L.dinv, L.S, L.Kel}) (void)hipFree(p);
        }
        for (double *p : {cg_r, cg_p, cg_w, cg_p2}) (void)hipFree(p);
        (void)hipFree(run_cnt);
        run_cnt = nullptr;
        (void)hipFree(run_ctl);
        run_ctl = nullptr;
        (void)hipFree(lan_ctl);
        lan_ctl = nullptr;
        coarse_direct_free();
        for (LanBuf &b : lan) {
            (void)hipFree(b.V);
            (void)hipFree(b.coef);
            (void)hipFree(b.part);
            (void)hipFree(b.ticket);
            (void)hipFree(b.mticket);
            (void)hipHostFree(b.hc);
            b = LanBuf();
        }
        for (int i = 0; i < LV_SLOTS; i++) {
            if (lan_graph[i]) (void)hipGraphExecDestroy(lan_graph[i]);
            lan_graph[i] = nullptr;
            lan_graph_state[i] = 0;
            if (lan_stream[i]) (void)hipStreamDestroy(lan_stream[i]);
            if (lan_done[i]) (void)hipEventDestroy(lan_done[i]);
            lan_stream[i] = nullptr;
            lan_done[i] = nullptr;
        }
        if (lan_fork) (void)hipEventDestroy(lan_fork);
        lan_fork = nullptr;
        for (int i = 0; i < LV_SLOTS; i++) {
            if (lv_ready[i]) (void)hipEventDestroy(lv_ready[i]);
            lv_ready[i] = nullptr;
            lv_ready_set[i] = false;
        }
        for (int i = 0; i < LV_SLOTS; i++) {
            if (pend_ev[i]) (void)hipEventDestroy(pend_ev[i]);
            pend_ev[i] = nullptr;
            pend[i].ptr = nullptr;
        }
    }
    // Lanczos work space per level (kept across design iterations): basis, coefficients, reduction partials, pinned
    // host copy of the coefficients; the runs of different levels are independent and may share the device
    struct LanBuf {
        double *V = nullptr, *coef = nullptr, *part = nullptr, *hc = nullptr;
        unsigned *ticket = nullptr, *mticket = nullptr;  // arrival counters of the chain's in-kernel reductions (its own: the chains of the levels run side by side)
        size_t cap = 0;
        int m = 0;
    };
    LanBuf lan[LV_SLOTS];
    // the run of a level is a static chain of ~400-1000 small launches: captured once into a hipGraph and replayed
    // every design iteration (host cost: one launch); rebuilt when the captured pointers may have changed
    hipGraphExec_t lan_graph[LV_SLOTS] = {};
    const void *lan_graph_key[LV_SLOTS][5] = {};
    long topology_epoch = 0;  // bumped by the owner whenever lists/buffers referenced by the operators are rebuilt
    int lan_graph_state[LV_SLOTS] = {};  // 0: not tried, 1: valid, -1: capture failed -> direct launches
    hipStream_t lan_stream[LV_SLOTS] = {};
    hipEvent_t lan_fork = nullptr, lan_done[LV_SLOTS] = {};
    // Round 6: the owner may say when a level's operator is complete (mark_level_ready, recorded on the solver's stream in the
    // middle of the assembly); that level's spectrum chain then waits for this event instead of for the end of the whole
    // assembly -- level 1's chain needs two small kernels, not the stencils of the levels below it.  Valid for one assembly.
    hipEvent_t lv_ready[LV_SLOTS] = {};
    bool lv_ready_set[LV_SLOTS] = {};
    int mark_level_ready(int l) {
        static const bool off = getenv("TP_NO_LEVEL_EVENTS") != nullptr;
        if (off || grid->has_comm || l < 0 || l >= LV_SLOTS) return TP_OK;
        if (!lv_ready[l]) TP_HIP(hipEventCreateWithFlags(&lv_ready[l], hipEventDisableTiming));
        TP_HIP(hipEventRecord(lv_ready[l], grid->stream));
        lv_ready_set[l] = true;
        return TP_OK;
    }

    // ---- operator application with one of the epilogues -------------------
    // out_halo: the ghost planes of a.out will be read next (another operator application, a grid transfer).  On the
    // tile levels the slab's boundary planes are then produced FIRST, their exchange starts on the second stream and
    // overlaps with the interior planes (halo() of the consumer waits for it); everywhere else out_halo is ignored
    // and the consumer's halo() exchanges as before.  Per-node arithmetic does not depend on the split: bitwise the
    // same result.
    template <int EPI>
    int op(int l, NodeArgs a, bool out_halo = false) {
        Level<DOF> &L = lv[l];
        const long nown = L.g.owned_nodes();
        const int nb = (int)((nown + BLK - 1) / BLK);
        double bytes = 0.0, flops = 0.0;
        last_nblocks = nb;
        if (pend[l].ptr && (pend[l].ptr == a.out || pend[l].ptr != a.x)) TP_TRY(drain_halo(l));  // stale / about to be overwritten
        const int n_bnd = (L.g.has_lo ? 1 : 0) + (L.g.has_hi ? 1 : 0);
        const bool tile_level = DOF == 3 && ((L.kind == LV_MATFREE && L.use_tile) || L.kind == LV_MACRO);
        const bool split = out_halo && tile_level && EPI != EPI_APPLY_DOT && EPI != EPI_CHEB_DOT && !L.no_comm && n_bnd > 0 &&
                           halo_can_overlap(grid) && (L.g.own_hi - L.g.own_lo + 1) > n_bnd;
        // the two passes of a split launch: boundary planes (one or two single-plane ranges), then the interior
        auto tile_ranges = [&](int pass, int &lo, int &hi, int &r1lo, int &r1hi) {
            lo = L.g.own_lo, hi = L.g.own_hi, r1lo = 0, r1hi = -1;
            if (!split) return;
            if (pass == 0) {
                if (L.g.has_lo && L.g.has_hi) lo = hi = L.g.own_lo, r1lo = r1hi = L.g.own_hi;
                else if (L.g.has_lo) hi = lo;
                else lo = hi;
            } else {
                lo += L.g.has_lo ? 1 : 0;
                hi -= L.g.has_hi ? 1 : 0;
            }
        };
        auto after_boundary = [&]() -> int {  // DMGlobalToLocalBegin on the output
            if (!pend_ev[l]) TP_HIP(hipEventCreateWithFlags(&pend_ev[l], hipEventDisableTiming));
            const int rc = halo_nodes_begin(grid, L.g, a.out, DOF, pend_ev[l]);
            if (rc == TP_OK) pend[l].ptr = a.out;
            return rc == 2 ? TP_OK : rc;  // 2: no in-place exchange after all -> the consumer's halo() does it
        };
        if (DOF == 3 && L.kind == LV_MATFREE && L.use_tile) {
            const int tx = (L.g.nx + TOUT - 1) / TOUT, ty = (L.g.ny + TOUT - 1) / TOUT;
            const int planes = L.g.own_hi - L.g.own_lo + 1;
            static const int kz_env = getenv("TP_TILE_KZ") ? atoi(getenv("TP_TILE_KZ")) : 0;
            static const int fine_v = fine_version();
            int kz = kz_env > 0 ? kz_env : fine_kz(planes, tx * ty, fine_v);
            // third generation (fine_u4.h; TP_FINE_V=2: k_fine_tile, 1: k_matfree_tile): tile shape by mesh size.  Its
            // 32-bit window arithmetic needs every vector of the level below 2 GB.
            const int t32 = ((L.g.nx + 30) / 31) * ((L.g.ny + 6) / 7);   // 32 x 8 tiles per z-chunk
            const int gen = fine_generation(L);
            if (a.pz && gen != 2) return TP_ERR_STATE;  // the fused p update exists in k_fine_tile only: never run it on a kernel that ignores it
            if (gen == 3) {
                constexpr bool IS_CHEB = (EPI == EPI_CHEB || EPI == EPI_CHEB_DOT);
                static const int shape_env = getenv("TP_FINE_SHAPE") ? atoi(getenv("TP_FINE_SHAPE")) : 0;  // 1: 16x16, 2: 32x8
                // measured on the BASELINE meshes (tools/probe/fine_probe.hip, profiles/r03_fine_probe.txt): the long
                // rows of 32 x 8 win once a chunk of them fills the chip (256^3, 512x256x256; Chebyshev from 256x128x128)
                const bool wide = shape_env ? shape_env >= 2 : t32 >= (IS_CHEB ? 160 : 256);
                const bool timed3 = grid->kt_on && IS_CHEB && !split && !sg_capturing;
                if (timed3) kernel_timer_mark(grid);
                for (int pass = 0; pass < (split ? 2 : 1); pass++) {
                    int lo, hi, r1lo, r1hi;
                    tile_ranges(pass, lo, hi, r1lo, r1hi);
                    const int pl = hi - lo + 1;
                    int kz3;
                    dim3 gdim;
                    if (wide) {
                        const int nch = kz_env > 0 ? (pl + kz_env - 1) / kz_env : (pl + 42) / 43;  // chunks of <= 43 planes, balanced
                        kz3 = (pl + nch - 1) / nch;
                    } else {
                        kz3 = kz_env > 0 ? kz_env : fine_kz(pl > 0 ? pl : 1, tx * ty, fine_v, IS_CHEB ? 512 : 768);
                        if (kz3 > pl && pl > 0) kz3 = pl;
                    }
                    const int tz = (hi - lo + kz3) / kz3 + (r1hi - r1lo + kz3) / kz3;
                    TileArgs ta{L.g.nx, L.g.ny, L.g.nzl, L.g.ex, L.g.ey, L.g.ezl, lo, hi, kz3,
                                L.E, L.mask, L.colmask, L.sym_slot * SYMKE_STRIDE, 0, 0, nullptr, xcd_remap(), 0, r1lo, r1hi,
                                0, nullptr, nullptr, 0, nullptr};
                    if (wide && shape_env == 3) {  // experiment: 32 x 16 threads (31 x 15 nodes out: 1.10 instead of 1.18 x the bytes), one workgroup of 8 waves per CU
                        gdim = dim3((L.g.nx + 30) / 31, (L.g.ny + 14) / 15, tz);
                        last_nblocks = gdim.x * gdim.y * gdim.z;
                        TP_LAUNCH((k_fine_u4<EPI, 32, 16, 1, true>), gdim, dim3(512), 0, grid->stream, ta, a);
                    } else if (wide) {
                        gdim = dim3((L.g.nx + 30) / 31, (L.g.ny + 6) / 7, tz);
                        last_nblocks = gdim.x * gdim.y * gdim.z;
                        TP_LAUNCH((k_fine_u4<EPI, 32, 8, 2, true>), gdim, dim3(256), 0, grid->stream, ta, a);
                    } else {
                        gdim = dim3(tx, ty, tz);
                        last_nblocks = gdim.x * gdim.y * gdim.z;
                        TP_LAUNCH((k_fine_u4<EPI, 16, 16, IS_CHEB ? 2 : 3, true>), gdim, dim3(256), 0, grid->stream, ta, a);
                    }
                    if (split && pass == 0) TP_TRY(after_boundary());
                }
                if (timed3) kernel_timer_mark(grid);
                bytes = 16.0 * DOF * nown + 8.0 * L.g.own_elems();
                flops = 2.0 * 576 * (double)L.g.own_elems();
            } else {
            if (kz > planes) kz = planes;
            const bool timed = grid->kt_on && (EPI == EPI_CHEB || EPI == EPI_CHEB_DOT) && !split && !sg_capturing;
            if (timed) kernel_timer_mark(grid);
            for (int pass = 0; pass < (split ? 2 : 1); pass++) {
                int lo, hi, r1lo, r1hi;
                tile_ranges(pass, lo, hi, r1lo, r1hi);
                const int tz = (hi - lo + kz) / kz + (r1hi - r1lo + kz) / kz;
                last_nblocks = tx * ty * tz;
                TileArgs ta{L.g.nx, L.g.ny, L.g.nzl, L.g.ex, L.g.ey, L.g.ezl, lo, hi, kz,
                            L.E, L.mask, L.colmask, L.sym_slot * SYMKE_STRIDE, 0, 0, nullptr, xcd_remap(), 0, r1lo, r1hi,
                            0, nullptr, nullptr, 0, nullptr};
                if (gen == 2) {
                    TP_LAUNCH((k_fine_tile<EPI>), dim3(tx, ty, tz), dim3(TILE * TILE), 0, grid->stream, ta, a);
                } else {
                    if constexpr (EPI == EPI_CHEB_DOT) return TP_ERR_STATE;
                    else TP_LAUNCH((k_matfree_tile<EPI, 0>), dim3(tx, ty, tz), dim3(TILE * TILE), 0, grid->stream, ta, a);
                }
                if (split && pass == 0) TP_TRY(after_boundary());
            }
            if (timed) kernel_timer_mark(grid);
            bytes = 16.0 * DOF * nown + 8.0 * L.g.own_elems();
            flops = 2.0 * 576 * (double)L.g.own_elems();
            }
        } else if constexpr (EPI == EPI_CHEB_DOT) {
            return TP_ERR_STATE;  // only the fine tile kernel carries the fused b . x_out
        } else if (DOF == 3 && L.kind == LV_MACRO) {
            const int tx = (L.g.nx + TOUT - 1) / TOUT, ty = (L.g.ny + TOUT - 1) / TOUT;
            const int planes = L.g.own_hi - L.g.own_lo + 1;
            static const int kz_env = getenv("TP_MACRO_KZ") ? atoi(getenv("TP_MACRO_KZ")) : 0;
            int kz = kz_env > 0 ? kz_env : (int)((long)planes * tx * ty / 768);
            if (kz_env <= 0) {
                // small levels: chunks short enough for about one workgroup per CU (64^3..192x64x64 elements: kz 1-2
                // instead of 4 is 3-9 % of the whole design iteration), never longer than 4 below one round of slots
                int kmin = (int)(((long)planes * tx * ty + 128) / 256);
                kmin = kmin < 1 ? 1 : (kmin > 4 ? 4 : kmin);
                kz = kz < kmin ? kmin : (kz > 64 ? 64 : kz);
            }
            if (kz > planes) kz = planes;
            // Dirichlet correction of the level-1 operator.  One launch computes tiles AND element-row products
            // (extra workgroups behind the tiles), a second one adds the gathered products to the result.  Slab runs keep
            // the older order (products first, added by the tiles): the boundary planes of a boundary-first launch
            // leave for the neighbour right after the first pass and must be final by then.
            static const bool no_fuse = getenv("TP_NO_CORR_FUSE") != nullptr;
            const bool fuse_corr = L.ncorr_nodes && !grid->has_comm && !no_fuse && EPI != EPI_APPLY_DOT;  // slabs: one order for both halo modes
            if (L.ncorr_nodes && !fuse_corr) {
                TP_LAUNCH(k_macro_corr_rows, dim3((L.nflag * 24 + BLK - 1) / BLK), dim3(BLK), 0, grid->stream, L.g,
                                   L.dK, L.flag_list, L.nflag, a.x, L.corr_tmp);
                TP_LAUNCH(k_macro_corr_gather, dim3((L.ncorr_nodes + BLK - 1) / BLK), dim3(BLK), 0, grid->stream,
                                   L.corr_nodes, L.corr_adj, L.ncorr_nodes, L.corr_tmp, L.corr, L.nflag);
                count_launch(grid);
                count_launch(grid);
            }
            for (int pass = 0; pass < (split ? 2 : 1); pass++) {
                int lo, hi, r1lo, r1hi;
                tile_ranges(pass, lo, hi, r1lo, r1hi);
                int tz = (hi - lo + kz) / kz + (r1hi - r1lo + kz) / kz;
                const int ntiles = tx * ty * tz;
                last_nblocks = ntiles;
                if (fuse_corr) tz += ((L.nflag * 24 + BLK - 1) / BLK + tx * ty - 1) / (tx * ty);  // row-product workgroups
                TileArgs ta{L.g.nx, L.g.ny, L.g.nzl, L.g.ex, L.g.ey, L.g.ezl, lo, hi, kz,
                            L.E, nullptr, nullptr, L.sym_slot * SYMKE_STRIDE, L.fex, L.fey,
                            (L.ncorr_nodes && !fuse_corr) ? L.corr : nullptr, xcd_remap(), L.sym_slot * MACG_STRIDE, r1lo, r1hi,
                            ntiles, L.dK, L.flag_list, L.nflag, L.corr_tmp};
                TP_LAUNCH((k_matfree_tile<EPI, 1>), dim3(tx, ty, tz), dim3(TILE * TILE), 0, grid->stream, ta, a);
                if (split && pass == 0) TP_TRY(after_boundary());
            }
            if (fuse_corr) {
                if constexpr (EPI != EPI_APPLY_DOT) {
                    TP_LAUNCH((k_macro_corr_apply<EPI>), dim3((L.ncorr_nodes + BLK - 1) / BLK), dim3(BLK), 0, grid->stream,
                              L.corr_nodes, L.corr_adj, L.ncorr_nodes, L.corr_tmp, L.nflag, a);
                }
                count_launch(grid);
            }
            bytes = 16.0 * DOF * nown + 8.0 * 8.0 * L.g.own_elems();
            flops = 2.0 * 576 * 8.0 * (double)L.g.own_elems();
        } else if (L.kind == LV_MATFREE) {
            static const bool no_st = getenv("TP_NO_PDE_STENCIL") != nullptr;
            bool as_stencil = false;
            if constexpr (DOF == 1) {
                if (L.wtab && !L.E && !L.mask && !no_st) {   // constant-coefficient scalar operator: its 27-point stencil form
                    ScalarStencilOp so{L.wtab, L.g};
                    TP_LAUNCH((k_node<1, ScalarStencilOp, EPI>), dim3(nb), dim3(BLK), 0, grid->stream, so, a);
                    bytes = 16.0 * nown;
                    flops = 2.0 * 27 * (double)nown;
                    as_stencil = true;
                }
            }
            if (!as_stencil) {
                MatfreeOp<DOF> o{L.KE, L.E, L.mask, L.g};
                TP_LAUNCH((k_node<DOF, MatfreeOp<DOF>, EPI>), dim3(nb), dim3(BLK), 0, grid->stream, o, a);
                bytes = 16.0 * DOF * nown + (L.E ? 8.0 * L.g.own_elems() : 0.0);
                flops = 2.0 * (8 * DOF) * (8 * DOF) * (double)L.g.own_elems();
            }
        } else {
            DiaOp<DOF> o{L.S, L.ndof(), L.g};
            const long rows_all = nown * DOF;
            // the row split (how many threads share a row) follows the size of the LEVEL, not of the launch: the boundary-first
            // launches below must sum every row in the order the single launch would
            const int nbr_all = (int)((rows_all + BLK - 1) / BLK);
            static const int split_env = getenv("TP_DIA_SPLIT") ? atoi(getenv("TP_DIA_SPLIT")) : -1;
            // DOF 3, round 6: the node form (k_dia_node3: a wave per z-offset, mirrored reads) serves EVERY level beyond the 9-way
            // class.  Measured per Chebyshev step against the unsplit row form the large levels used to run: C3's level 2 (70 785
            // nodes) 27.9 -> 19.0 us, the 256^3 class's (274 625) 121.5 -> 73.2, C5's (545 025) 231.7 -> 158.2 -- the mirrored reads
            // halve the coefficient stream out of HBM (the second use comes from the L2 / Infinity Cache); an unsplit node form
            // (one thread over all 27 neighbours) measured 209 / 387 us: too many registers per thread to keep the stream busy.
            static const bool by_node = !(getenv("TP_DIA_NODE") && atoi(getenv("TP_DIA_NODE")) == 0);
            const int rsplit = split_env >= 0 ? split_env : (nbr_all < 128 ? 9 : ((DOF == 3 && by_node) ? 3 : (nbr_all < 512 ? 3 : 1)));
            static const bool sym = getenv("TP_NO_DIA_SYM") == nullptr;
            auto launch_rows = [&](long t0, long tn, long t1, long tn1) -> int {
                o.t0 = t0, o.tn = tn, o.t1 = t1, o.tn1 = tn1;
                const long rows = tn < 0 ? rows_all : tn + tn1;
                int nbr = 0;
                if (rsplit == 9) {
                    nbr = (int)((rows + BLK / 9 - 1) / (BLK / 9));
                    TP_LAUNCH((k_dia_row_split<DOF, EPI, 9>), dim3(nbr), dim3(BLK), 0, grid->stream, o, a);
                } else if (rsplit == 3) {
                    // round 6: a thread per node and z-offset (k_dia_node3: the same bits, a third of the waves); TP_DIA_NODE=0: per row
                    bool done = false;
                    if constexpr (DOF == 3) {
                        if (by_node && EPI != EPI_APPLY_DOT) {
                            nbr = (int)((rows / 3 + 63) / 64);
                            if (sym)
                                TP_LAUNCH((k_dia_node3<EPI, true>), dim3(nbr), dim3(192), 0, grid->stream, o, a);
                            else
                                TP_LAUNCH((k_dia_node3<EPI, false>), dim3(nbr), dim3(192), 0, grid->stream, o, a);
                            done = true;
                        }
                    }
                    if (!done) {
                        nbr = (int)((rows + BLK / 3 - 1) / (BLK / 3));
                        if (sym)
                            TP_LAUNCH((k_dia_row_split<DOF, EPI, 3, true>), dim3(nbr), dim3(BLK), 0, grid->stream, o, a);
                        else
                            TP_LAUNCH((k_dia_row_split<DOF, EPI, 3>), dim3(nbr), dim3(BLK), 0, grid->stream, o, a);
                    }
                } else {
                    nbr = (int)((rows + BLK - 1) / BLK);
                    TP_LAUNCH((k_dia_row<DOF, EPI>), dim3(nbr), dim3(BLK), 0, grid->stream, o, a);
                }
                last_nblocks = nbr;
                return TP_OK;
            };
            // Halo overlap on the stencil levels (round 5; slabs): the rows of the one or two boundary planes in ONE launch
            // first, their exchange on the second stream, then the interior rows -- as on the tile levels, bitwise the same
            // result (a row's sum does not depend on the launch it is computed in).  TP_STENCIL_OVERLAP=0: one launch, the
            // consumer's halo() exchanges.
            static const bool st_ovl = !(getenv("TP_STENCIL_OVERLAP") && atoi(getenv("TP_STENCIL_OVERLAP")) == 0);
            const int planes = L.g.own_hi - L.g.own_lo + 1;
            const bool split_dia = st_ovl && out_halo && EPI != EPI_APPLY_DOT && !L.no_comm && n_bnd > 0 && halo_can_overlap(grid) &&
                                   planes > n_bnd && !sg_capturing;
            if (split_dia) {
                const long pr = (long)DOF * L.g.plane();
                const bool both = L.g.has_lo && L.g.has_hi;
                TP_TRY(launch_rows(L.g.has_lo ? 0 : rows_all - pr, pr, both ? rows_all - pr : 0, both ? pr : 0));
                TP_TRY(after_boundary());
                TP_TRY(launch_rows(L.g.has_lo ? pr : 0, rows_all - pr * n_bnd, 0, 0));
                grid->launches++;
            } else {
                TP_TRY(launch_rows(0, -1, 0, 0));
            }
            bytes = (27.0 * DOF * DOF + 2.0 * DOF) * 8.0 * nown;
            flops = 2.0 * 27 * DOF * DOF * (double)nown;
        }
        if (EPI == EPI_RESID) bytes += 8.0 * DOF * nown;
        // d (r/w), b, dinv -- the fine tile kernel instead reads b and the previous iterate (3-term form, diagonal on the fly)
        // (3-term form: the first step of a sweep -- c1 = 0 or the zero guess -- does not read a previous iterate)
        if (EPI == EPI_CHEB || EPI == EPI_CHEB_DOT) bytes += (three_term(L) ? ((a.c1 != 0.0 && !a.prev_zero) ? 2.0 : 1.0) : 4.0) * 8.0 * DOF * nown;
        if (grid->kt_on && l == 0 && (EPI == EPI_CHEB || EPI == EPI_CHEB_DOT) && !split && !sg_capturing) grid->kt_bytes += bytes;
        count_launch(grid, bytes, flops);
        if (split) grid->launches++;
        return TP_OK;
    }
    // ---- halos in flight on the second stream (one per level: the output of the last split launch)
    struct PendingHalo {
        const double *ptr = nullptr;
    };
    PendingHalo pend[LV_SLOTS];
    hipEvent_t pend_ev[LV_SLOTS] = {};
    int drain_halo(int l) {  // DMGlobalToLocalEnd
        if (pend[l].ptr) {
            TP_HIP(hipStreamWaitEvent(grid->stream, pend_ev[l], 0));
            pend[l].ptr = nullptr;
        }
        return TP_OK;
    }
    int drain_halos() {
        for (int l = 0; l < LV_SLOTS; l++) TP_TRY(drain_halo(l));
        return TP_OK;
    }
    // A kernel that writes whole owned planes of `out` one node at a time (vector updates, grid transfers), issued so
    // that the ghost planes of `out` travel while its interior is computed: launch(p0, np) covers the owned planes
    // [p0, p0 + np).  Tile levels of a slab run with the overlap available: boundary plane(s) first, exchange started
    // on the second stream, then the interior; everywhere else one launch over all owned planes (the consumer's
    // halo() exchanges).
    template <class F>
    int planes_split(int l, double *out, F launch) {
        Level<DOF> &L = lv[l];
        const int lo = L.g.own_lo, hi = L.g.own_hi;
        const int n_bnd = (L.g.has_lo ? 1 : 0) + (L.g.has_hi ? 1 : 0);
        const bool tile_level = DOF == 3 && ((L.kind == LV_MATFREE && L.use_tile) || L.kind == LV_MACRO);
        if (pend[l].ptr) TP_TRY(drain_halo(l));
        if (!(tile_level && !L.no_comm && n_bnd > 0 && halo_can_overlap(grid) && hi - lo + 1 > n_bnd)) return launch(lo, hi - lo + 1);
        if (L.g.has_lo) TP_TRY(launch(lo, 1));
        if (L.g.has_hi) TP_TRY(launch(hi, 1));
        if (!pend_ev[l]) TP_HIP(hipEventCreateWithFlags(&pend_ev[l], hipEventDisableTiming));
        const int rc = halo_nodes_begin(grid, L.g, out, DOF, pend_ev[l]);
        if (rc == TP_OK) pend[l].ptr = out;
        else if (rc != 2) return rc;
        return launch(lo + (L.g.has_lo ? 1 : 0), hi - lo + 1 - n_bnd);
    }
    // Fine tile kernel: Chebyshev in its 3-term form  u+ = u + c1 (u - u-) + c2 D^-1 (b - A u); u- sits in the output
    // buffer (read and overwritten by the same thread), so no direction vector is streamed.
    static bool three_term(const Level<DOF> &L) { return DOF == 3 && L.kind == LV_MATFREE && L.use_tile; }
    // Which kernel generation serves a tuned matrix-free level?  ONE place decides, op<>() launches what it says and solve()
    // asks it before it hands the fused p update to the product (only k_fine_tile takes NodeArgs::pz / pnew: ADVICE r4).
    // 3: fine_u4.h (its 32-bit window arithmetic needs every vector of the level below 2 GB), 2: fine_tile.h, 1: matfree_tile.h
    static int fine_generation(const Level<DOF> &L) {
        if (!(DOF == 3 && L.kind == LV_MATFREE && L.use_tile)) return 0;
        const int fine_v = fine_version();
        const int t32 = ((L.g.nx + 30) / 31) * ((L.g.ny + 6) / 7);   // 32 x 8 tiles per z-chunk
        if ((fine_v >= 3 || (fine_v == 0 && t32 >= 160)) && 24.0 * L.g.nodes() < 2.0e9) return 3;
        return (fine_v >= 2 || fine_v == 0) ? 2 : 1;
    }
    static bool runs_fine_tile(const Level<DOF> &L) { return fine_generation(L) == 2; }
    int halo(int l, double *v) {
        if (lv[l].no_comm) return TP_OK;
        if (pend[l].ptr == v) return drain_halo(l);  // already under way: ordered behind it, nothing to exchange
        return halo_nodes(grid, lv[l].g, v, DOF);
    }

    // y = A_l u (ghost planes of u refreshed first)
    int apply(int l, double *u, double *y) {
        TP_TRY(halo(l, u));
        NodeArgs a{};
        a.x = u;
        a.out = y;
        return op<EPI_APPLY>(l, a);
    }
    // y = A_0 u with the operator of the Krylov method (the kernel of CG's A p; its dot product u . A u is discarded)
    int apply_krylov(double *u, double *y) {
        TP_TRY(halo(0, u));
        NodeArgs a{};
        a.x = u;
        a.out = y;
        a.partials = grid->partials;
        a.ticket = tail_ticket(grid);
        a.red_out = grid->scal + S_PW;
        TP_TRY(op<EPI_APPLY_DOT>(0, a));
        return finish_tail<1>(grid, last_nblocks, S_PW);
    }

    // ---- the long smoothing run of the coarsest level (30 steps of 4-5 us kernels) as a hipGraph: captured when its
    // arguments change -- the Chebyshev window once per design iteration, the x/x2 roles alternate between consecutive
    // V-cycles (odd number of steps) -- and replayed for the other V-cycles of the solve.  Launches per design
    // iteration at 128^3: 1660 -> 1016.  Time: 30.93 against 30.82 ms (three runs each, +-0.05): the two captures and
    // instantiations per design iteration cost what the saved host launches bring -- outside a profiler the host
    // keeps up with these kernels, the device does not wait for it.  Hence opt-in (TP_SMOOTH_GRAPH=1), kept as the
    // evidence for that statement.
    struct SmoothGraph {
        hipGraphExec_t exec = nullptr;
        const void *ptr[4] = {nullptr, nullptr, nullptr, nullptr};  // b, x, x2, d
        double theta = 0.0, delta = 0.0;
        int k = 0, flags = 0, level = -1;
        bool swap = false;      // the run leaves x and x2 exchanged
        double bytes = 0.0, flops = 0.0;  // accounting of one run
        long stamp = 0;
    };
    SmoothGraph sgraph[4];
    bool sg_capturing = false;
    hipStream_t sg_stream = nullptr;
    long sg_clock = 0;
    static bool smooth_graphs_on() {  // opt-in (TP_SMOOTH_GRAPH=1): measured 0.3 % SLOWER at 128^3, see the comment above
        const char *e = getenv("TP_SMOOTH_GRAPH");
        return e && atoi(e) != 0 && !tp_debug_sync();
    }
    void smooth_graphs_free() {
        for (SmoothGraph &g : sgraph) {
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
            g = SmoothGraph();
        }
        if (sg_stream) (void)hipStreamDestroy(sg_stream);
        sg_stream = nullptr;
    }
    int smooth_replay(int l, const double *b, int k, bool zero_guess, bool first_done, double theta, double delta) {
        Level<DOF> &L = lv[l];
        hipStream_t s = grid->stream;
        const int flags = (zero_guess ? 1 : 0) | (first_done ? 2 : 0);
        SmoothGraph *hit = nullptr, *victim = &sgraph[0];
        for (SmoothGraph &g : sgraph) {
            if (g.exec && g.level == l && g.ptr[0] == b && g.ptr[1] == L.x && g.ptr[2] == L.x2 && g.ptr[3] == L.d && g.theta == theta &&
                g.delta == delta && g.k == k && g.flags == flags)
                hit = &g;
            if (g.stamp < victim->stamp) victim = &g;
        }
        if (getenv("TP_DEBUG_GRAPH")) fprintf(stderr, "smooth graph: level %d %s\n", l, hit ? "hit" : "miss");
        if (hit && hipGraphLaunch(hit->exec, s) == hipSuccess) {
            hit->stamp = ++sg_clock;
            if (hit->swap) std::swap(L.x, L.x2);
            grid->launches += 1;
            grid->alg_bytes += hit->bytes;
            grid->flops += hit->flops;
            return TP_OK;
        }
        if (hit) {  // a replay that failed: drop the graph, run the launches
            (void)hipGetLastError();
            (void)hipGraphExecDestroy(hit->exec);
            *hit = SmoothGraph();
        }
        // first use of this argument set: run it directly now, capture the identical run for the next time
        double *const xa = L.x, *const xb = L.x2;  // roles before the run
        sg_capturing = true;
        int rc = smooth(l, b, k, zero_guess, -1, first_done);
        if (rc) {
            sg_capturing = false;
            return rc;
        }
        const bool swapped = L.x != xa;
        if (victim->exec) (void)hipGraphExecDestroy(victim->exec);
        *victim = SmoothGraph();
        // captured on a stream of our own (the grid's stream may be the legacy default stream, which cannot capture);
        // the graph is replayed on the grid's stream
        if (!sg_stream && hipStreamCreateWithFlags(&sg_stream, hipStreamNonBlocking) != hipSuccess) sg_stream = nullptr;
        if (!sg_stream || hipStreamBeginCapture(sg_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            sg_capturing = false;
            return TP_OK;
        }
        const long l0 = grid->launches;
        const double b0 = grid->alg_bytes, f0 = grid->flops;
        L.x = xa;  // the captured run starts from the same roles and leaves them like the direct run did
        L.x2 = xb;
        grid->stream = sg_stream;
        rc = smooth(l, b, k, zero_guess, -1, first_done);
        grid->stream = s;
        hipGraph_t g = nullptr;
        const hipError_t e1 = hipStreamEndCapture(sg_stream, &g);
        SmoothGraph ng;
        ng.bytes = grid->alg_bytes - b0;
        ng.flops = grid->flops - f0;
        grid->launches = l0;  // the captured chain was not executed
        grid->alg_bytes = b0;
        grid->flops = f0;
        sg_capturing = false;
        L.x = swapped ? xb : xa;
        L.x2 = swapped ? xa : xb;
        if (rc || e1 != hipSuccess || !g || hipGraphInstantiate(&ng.exec, g, nullptr, nullptr, 0) != hipSuccess) {
            if (getenv("TP_DEBUG_GRAPH")) fprintf(stderr, "smooth graph: capture failed rc=%d e1=%d g=%p\n", rc, (int)e1, (void *)g);
            (void)hipGetLastError();
            if (g) (void)hipGraphDestroy(g);
            return TP_OK;  // the direct run above did the work
        }
        if (getenv("TP_DEBUG_GRAPH")) fprintf(stderr, "smooth graph: captured level %d k %d swap %d\n", l, k, (int)swapped);
        (void)hipGraphDestroy(g);
        ng.ptr[0] = b;
        ng.ptr[1] = xa;
        ng.ptr[2] = xb;
        ng.ptr[3] = L.d;
        ng.theta = theta;
        ng.delta = delta;
        ng.k = k;
        ng.flags = flags;
        ng.level = l;
        ng.swap = swapped;
        ng.stamp = ++sg_clock;
        *victim = ng;
        return TP_OK;
    }
    // ---- the coarsest level solved exactly (coarse_direct.h): opt.coarse_direct, one rank or the replicated copy
    struct CoarseDirect {
        CdGeom g{};
        double *Lb = nullptr, *Tm = nullptr, *Ld = nullptr, *Linv = nullptr, *W = nullptr, *Wt = nullptr, *y = nullptr;
        XcdRunCtrl *ctl = nullptr;
        int level = -1;       // the level the factor belongs to (nlv - 1, or nlv: the replicated copy)
        bool factored = false;
    } cd;
    int cd_level() const { return replicate ? rix(nlv - 1) : nlv - 1; }
    bool cd_early = false;              // this assembly's factorisation is already under way (coarse_direct_early)
    bool cd_inverse_owed = false;       // ... and its triangular inverse is still to be enqueued behind it (estimate_spectra)
    int enqueue_owed_inverse() {
        if (!cd_inverse_owed) return TP_OK;
        cd_inverse_owed = false;
        const int l = nlv - 1;
        hipStream_t main = grid->stream;
        grid->stream = lan_stream[l];
        const int rc = coarse_direct_factor(2);
        grid->stream = main;
        if (rc) return rc;
        TP_HIP(hipEventRecord(lan_done[l], lan_stream[l]));
        return TP_OK;
    }
    hipStream_t side_stream = nullptr;  // owner's spare stream (idle during estimate_spectra): a second one for the chains
    // Called by the owner as soon as the coarsest level's stencil is enqueued (before the other levels are finished): the
    // factorisation goes to the coarsest level's stream right away.  One rank only (the replicated copy of a multi-rank
    // run is built later, setup_replicated); estimate_spectra then leaves the level alone.
    int coarse_direct_early(bool *started) {
        *started = false;
        static const bool serial = getenv("TP_LANCZOS_SERIAL") != nullptr;
        if (grid->has_comm || serial || opt.ksp_mode != 0 || nlv < 3 || !coarse_direct_ok()) return TP_OK;
        const int l = nlv - 1;
        hipStream_t main = grid->stream;
        if (!lan_fork) TP_HIP(hipEventCreateWithFlags(&lan_fork, hipEventDisableTiming));
        if (!lan_stream[l]) TP_HIP(hipStreamCreateWithFlags(&lan_stream[l], hipStreamNonBlocking));
        if (!lan_done[l]) TP_HIP(hipEventCreateWithFlags(&lan_done[l], hipEventDisableTiming));
        TP_HIP(hipEventRecord(lan_fork, main));
        TP_HIP(hipStreamWaitEvent(lan_stream[l], lan_fork, 0));
        // Only the fill and the factorisation itself now: the inverse's 17 launches follow from estimate_spectra, once the
        // rest of the assembly and the spectra chains are enqueued -- they are not needed for 1.4 ms, and enqueueing them here
        // kept the solver's stream idle for their host time in the middle of the assembly (round 6).
        static const bool split_enq = !(getenv("TP_CD_SPLIT_ENQUEUE") && atoi(getenv("TP_CD_SPLIT_ENQUEUE")) == 0);
        grid->stream = lan_stream[l];
        const int rc = coarse_direct_factor(split_enq ? 1 : 3);
        grid->stream = main;
        if (rc) return rc;
        cd_inverse_owed = split_enq;
        if (!split_enq) TP_HIP(hipEventRecord(lan_done[l], lan_stream[l]));
        cd_early = true;
        *started = true;
        return TP_OK;
    }
    bool coarse_direct_ok() const {
        if (!opt.coarse_direct || getenv("TP_NO_COARSE_DIRECT") || tp_xcd_disabled() || nlv < 2 || DOF != 3) return false;
        const Level<DOF> &L = lv[cd_level()];
        if (L.kind != LV_DIA || (grid->has_comm && !L.no_comm) || L.own_n() != L.ndof() || L.ndof() > CD_MAXROWS) return false;
        const long hb = (long)DOF * (L.g.plane() + L.g.nx + 1) + DOF - 1;
        const int KB = (int)((hb + CD_NB - 1) / CD_NB);
        if (!(KB >= 1 && KB <= CD_KBMAX && L.ndof() >= 4 * CD_NB)) return false;
        // a level of <= 448 rows runs its Chebyshev steps inside ONE workgroup at 0.4 us each (coarse_run.h): a
        // factorisation per assembly does not pay there -- coarse_direct = 2 asks for it anyway
        return opt.coarse_direct >= 2 || L.ndof() > (long)RUN_RPB * 8;
    }
    void coarse_direct_free() {
        if (cd_pending && lan_stream[nlv - 1]) (void)hipStreamSynchronize(lan_stream[nlv - 1]);
        cd_pending = false;
        for (double **p : {&cd.Lb, &cd.Tm, &cd.Ld, &cd.Linv, &cd.W, &cd.Wt, &cd.y}) {
            (void)hipFree(*p);
            *p = nullptr;
        }
        (void)hipFree(cd.ctl);
        cd.ctl = nullptr;
        cd.factored = false;
        cd.level = -1;
    }
    // factor + invert on grid->stream (the caller puts it on a stream of its own beside the spectra chains)
    // parts: 1 = band fill + factorisation, 2 = the triangular inverse behind it, 3 = both
    int coarse_direct_factor(int parts = 3) {
        const int l = cd_level();
        Level<DOF> &L = lv[l];
        hipStream_t s = grid->stream;
        CdGeom g;
        g.n = (int)L.ndof();
        g.np = (g.n + CD_NB - 1) / CD_NB * CD_NB;
        g.nblk = g.np / CD_NB;
        g.KB = (int)(((long)DOF * (L.g.plane() + L.g.nx + 1) + DOF - 1 + CD_NB - 1) / CD_NB);
        if (cd.level != l || cd.g.np != g.np || cd.g.KB != g.KB) {
            coarse_direct_free();
            TP_HIP(hipMalloc((void **)&cd.Lb, sizeof(double) * (size_t)g.nblk * (g.KB + 1) * CD_NB * CD_NB));
            TP_HIP(hipMalloc((void **)&cd.Ld, sizeof(double) * (size_t)g.nblk * CD_NB * CD_NB));
            TP_HIP(hipMalloc((void **)&cd.Linv, sizeof(double) * (size_t)g.nblk * CD_NB * CD_NB));
            TP_HIP(hipMalloc((void **)&cd.W, sizeof(double) * (size_t)g.np * g.np));
            TP_HIP(hipMalloc((void **)&cd.Tm, sizeof(double) * (size_t)g.np * g.np));
            TP_HIP(hipMalloc((void **)&cd.Wt, sizeof(double) * (size_t)g.np * g.np));
            TP_HIP(hipMalloc((void **)&cd.y, sizeof(double) * (size_t)g.np));
            TP_HIP(hipMalloc((void **)&cd.ctl, sizeof(XcdRunCtrl)));
            TP_HIP(hipMemsetAsync(cd.ctl, 0, sizeof(XcdRunCtrl), s));
            cd.level = l;
        }
        cd.g = g;
        const int P = g.KB + 1;
        if (parts & 1) {
        TP_HIP(hipMemsetAsync(cd.Lb, 0, sizeof(double) * (size_t)g.nblk * (g.KB + 1) * CD_NB * CD_NB, s));
        DiaOp<DOF> o{L.S, L.ndof(), L.g};
        TP_LAUNCH((k_cd_fill<DOF>), dim3((g.np + CD_T - 1) / CD_T), dim3(CD_T), 0, s, o, g, cd.Lb);
        static const int stages = getenv("TP_CD_STAGES") ? atoi(getenv("TP_CD_STAGES")) : 3;  // (timing aid: 1 fill, 2 + factor, 3 all)
        static const bool prof_on = getenv("TP_CD_PROF") != nullptr;  // (timing aid: ticks per phase, printed per factorisation)
        long long *prof = nullptr;
        if (prof_on) {
            TP_HIP(hipMalloc((void **)&prof, sizeof(long long) * 8 * 32));
            TP_HIP(hipMemsetAsync(prof, 0, sizeof(long long) * 8 * 32, s));
        }
        if (stages >= 2) TP_LAUNCH(k_cd_factor, dim3(8 * P), dim3(CD_T), 0, s, g, cd.Lb, cd.Ld, cd.ctl, P, prof);
        if (prof_on) {
            long long h[8 * 32];
            TP_HIP(hipStreamSynchronize(s));
            TP_HIP(hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost));
            (void)hipFree(prof);
            int rate = 100000;
            (void)hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
            for (int r = 0; r < P; r += (P > 4 ? P / 3 : 1))
                fprintf(stderr, "cd factor rank %2d us: A %.0f | B diag %.0f | B look-ahead %.0f | barrier1 %.0f | C %.0f | barrier2 %.0f | loop %.0f\n", r,
                        h[r * 8 + 0] * 1e3 / rate, h[r * 8 + 1] * 1e3 / rate, h[r * 8 + 2] * 1e3 / rate, h[r * 8 + 3] * 1e3 / rate, h[r * 8 + 4] * 1e3 / rate,
                        h[r * 8 + 5] * 1e3 / rate, h[r * 8 + 7] * 1e3 / rate);
        }
        }
        if (!(parts & 2)) return TP_OK;
        static const int stages2 = getenv("TP_CD_STAGES") ? atoi(getenv("TP_CD_STAGES")) : 3;
        if (stages2 >= 3) {
            TP_LAUNCH(k_cd_diag_inv, dim3(g.nblk), dim3(WAVE), 0, s, cd.Ld, cd.Linv);
            static const bool dc = getenv("TP_CD_INVERT_COLUMNS") == nullptr;  // (1: round 3's block-column substitution)
            if (dc) {
                TP_LAUNCH(k_cd_dc_diag, dim3(g.nblk), dim3(CD_T), 0, s, g, cd.Linv, cd.W, cd.Wt);
                for (int lv = 1; (1 << (lv - 1)) < g.nblk; lv++) {
                    const int half = 1 << (lv - 1), nseg = (g.nblk + 2 * half - 1) / (2 * half);
                    TP_LAUNCH(k_cd_dc_t, dim3(half, std::min(g.KB, half), nseg), dim3(WAVE), 0, s, g, lv, cd.Lb, cd.W, cd.Tm);
                    TP_LAUNCH(k_cd_dc_w, dim3(half, half, nseg), dim3(WAVE), 0, s, g, lv, cd.Tm, cd.W, cd.Wt);
                    grid->launches += 2;
                }
            } else {
                TP_LAUNCH(k_cd_invert, dim3(g.nblk), dim3(CD_T), 0, s, g, cd.Lb, cd.Linv, cd.W, cd.Wt);
            }
        }
        grid->launches += 4;
        const double nb2 = (double)g.np * g.np;
        grid->alg_bytes += 8.0 * (27.0 * DOF * DOF * L.g.nodes() + nb2);  // stencil in, W and W^T (lower halves) out
        grid->flops += (double)g.np * g.KB * CD_NB * (g.KB * CD_NB + g.np);  // band Cholesky + triangular inverse
        cd.factored = true;
        return TP_OK;
    }
    // the factorisation enqueued by this assembly is still running on its side stream: the solver's stream waits for it
    // (device-side; the host does not block)
    bool cd_pending = false;
    bool cd_deferred_last = false;  // the last assembly's factorisation ran beside the head of the solve (give-up severity)
    int join_pending_factor() {
        if (cd_pending) {
            cd_pending = false;
            TP_HIP(hipStreamWaitEvent(grid->stream, lan_done[nlv - 1], 0));
        }
        return TP_OK;
    }
    // x = A^-1 b on level cd.level
    int coarse_direct_apply(int l, const double *b) {
        Level<DOF> &L = lv[l];
        TP_TRY(join_pending_factor());
        const int rows_per = CD_T / WAVE, nb = (cd.g.n + rows_per - 1) / rows_per;
        TP_LAUNCH(k_cd_tri<false>, dim3(nb), dim3(CD_T), 0, grid->stream, cd.g, cd.W, b, cd.y);
        TP_LAUNCH(k_cd_tri<true>, dim3(nb), dim3(CD_T), 0, grid->stream, cd.g, cd.Wt, cd.y, L.x);
        grid->launches += 2;
        grid->alg_bytes += 8.0 * ((double)cd.g.n * cd.g.n + 4.0 * cd.g.n);
        grid->flops += 2.0 * (double)cd.g.n * cd.g.n;
        return TP_OK;
    }
    // ---- the coarsest level's run in one launch (coarse_run.h)
    unsigned long long *run_cnt = nullptr;  // [dev] arrival counter (monotone over the runs) + give-up flag
    XcdRunCtrl *run_ctl = nullptr;          // [dev] control block of the one-XCD run (zero between runs)
    bool gaveup_seen = false;               // the last diverged solve found a give-up flag raised by a one-XCD kernel
    // did a one-XCD kernel (Chebyshev run, Lanczos run, factorisation) give up?  Blocking read of the sticky flags.
    bool xcd_gaveup() {
        (void)join_pending_factor();
        unsigned long long f[3] = {0ull, 0ull, 0ull};
        XcdRunCtrl *blocks[3] = {run_ctl, lan_ctl, cd.ctl};
        for (int q = 0; q < 3; q++)
            if (blocks[q]) (void)hipMemcpyAsync(&f[q], &blocks[q]->gaveup[0], sizeof(unsigned long long), hipMemcpyDeviceToHost, grid->stream);
        (void)hipStreamSynchronize(grid->stream);
        gaveup_mask = (f[0] ? 1 : 0) | (f[1] ? 2 : 0) | (f[2] ? 4 : 0);
        return gaveup_mask != 0;
    }
    int gaveup_mask = 0;  // which control block raised the flag xcd_gaveup() found: 1 Chebyshev run, 2 Lanczos run, 4 factorisation
    void xcd_reset_controls() {
        for (XcdRunCtrl *b : {run_ctl, lan_ctl, cd.ctl})
            if (b) (void)hipMemsetAsync(b, 0, sizeof(XcdRunCtrl), grid->stream);
        cd_early = cd_inverse_owed = false;
        cd.factored = false;
    }
    // after an assembly that failed half way: no chain of a side stream may still be running when the next one starts
    void join_side_streams() {
        for (int i = 0; i < LV_SLOTS; i++)
            if (lan_stream[i]) (void)hipStreamSynchronize(lan_stream[i]);
        if (side_stream) (void)hipStreamSynchronize(side_stream);
        cd_early = cd_inverse_owed = false;
        cd_pending = false;
    }
    unsigned long long run_base = 0;        // arrivals of all runs so far
    long coarse_runs = 0;
    // rows per thread: the fewest that bring the run down to `want` workgroups (barrier cost grows with their number)
    static int run_rows_per_thread(long rows, int *wgs) {
        static const int want = getenv("TP_RUN_WGS") ? atoi(getenv("TP_RUN_WGS")) : 16;
        int R = 1;
        while (R < 8 && (rows + (long)RUN_RPB * R - 1) / ((long)RUN_RPB * R) > want) R *= 2;
        *wgs = (int)((rows + (long)RUN_RPB * R - 1) / ((long)RUN_RPB * R));
        return R;
    }
    static int xcd_rows_per_thread(long rows, int *wgs) {  // as few rows per thread as 32 workgroups allow
        int R = 1;
        while (R < 8 && (rows + (long)RUN_RPB * R - 1) / ((long)RUN_RPB * R) > 32) R *= 2;
        *wgs = (int)((rows + (long)RUN_RPB * R - 1) / ((long)RUN_RPB * R));
        return R;
    }
    // 3: one launch whose workgroups all sit on ONE XCD and exchange the iterate through its L2 (coarse_run.h; on for
    //    449 .. 14336 rows held by one rank unless TP_NO_COARSE_XCD / TP_NO_COARSE_RUN: 2.3 us per step against 3-4 per launch)
    // 0: separate launches; 1: one launch of ONE workgroup (iterate in LDS; on unless TP_NO_COARSE_RUN);
    // 2: one launch of several workgroups with a barrier per step (opt-in TP_COARSE_RUN=1: measured at 128^3 / C1 / C3 it
    // costs what its launches cost, 19.7 against 19.6 ms at 654 against 1471 launches per design iteration -- a step inside
    // the kernel is 2.6-3.2 us (tools/probe/step_probe.hip), a dependent launch 3.1 us: the XCDs' L2 slices are not coherent,
    // either way the iterate makes a round trip through the memory side; and a spinning kernel is a liability on a shared
    // device)
    int coarse_run_mode(int l, int nsteps) const {
        const Level<DOF> &L = lv[l];
        if (sg_capturing || DOF != 3 || L.kind != LV_DIA || !coarsest(l)) return 0;
        if (!(L.no_comm || !grid->has_comm) || nsteps < 4 || nsteps > RUN_MAXK) return 0;
        int wgs;
        const int R = run_rows_per_thread(L.own_n(), &wgs);
        if (L.own_n() <= (long)RUN_RPB * 8 && L.ndof() <= RUN_XS && L.own_n() == L.ndof()) return getenv("TP_NO_COARSE_RUN") ? 0 : 1;
        const char *sw = getenv("TP_COARSE_RUN");  // read per call: the tests switch it within a process
        if (sw && atoi(sw) == 1) return wgs <= RUN_MAX_WGS && run_stage_doubles(L.g, DOF, R) <= RUN_XS ? 2 : 0;
        // 3: the run on one XCD (coarse_run.h): one rank, at most 32 workgroups (one per CU of an XCD), R <= 2
        if (getenv("TP_NO_COARSE_RUN") || getenv("TP_NO_COARSE_XCD") || tp_xcd_disabled()) return 0;
        return xcd_eligible(l, RUN_XS, 8) ? 3 : 0;
    }
    // the level fits a run on one XCD: all its rows on this rank (one rank, or the replicated copy of the coarsest
    // level), 2 .. 32 workgroups (one per CU of an XCD) of at most max_r rows per thread
    bool xcd_eligible(int l, long stage_cap, int max_r) const {
        const Level<DOF> &L = lv[l];
        if (sg_capturing || DOF != 3 || L.kind != LV_DIA || (grid->has_comm && !L.no_comm) || tp_debug_sync()) return false;
        int wx;
        const int Rx = xcd_rows_per_thread(L.own_n(), &wx);
        return Rx <= max_r && wx <= 32 && wx >= 2 && run_stage_doubles(L.g, DOF, Rx) <= stage_cap && L.own_n() == L.ndof();
    }
    // steps it0 .. k-1 of smooth() (it0 >= 1: the direction vector L.d is valid)
    int coarse_run(int l, const double *b, int it0, int k, double sigma, double delta, int mode) {
        Level<DOF> &L = lv[l];
        if (!run_cnt) {
            TP_HIP(hipMalloc((void **)&run_cnt, 2 * sizeof(unsigned long long)));
            TP_HIP(hipMemset(run_cnt, 0, 2 * sizeof(unsigned long long)));
        }
        ChebRunCoef cr;
        cr.nsteps = k - it0;
        double rho = 1.0 / sigma;
        for (int s = 0; s < cr.nsteps; s++) {
            const double rn = 1.0 / (2.0 * sigma - rho);
            cr.c1[s] = rn * rho;
            cr.c2[s] = 2.0 * rn / delta;
            rho = rn;
        }
        DiaOp<DOF> o{L.S, L.ndof(), L.g};
        if (mode == 1) {
            TP_LAUNCH((k_dia_cheb_run<DOF, 8, true>), dim3(1), dim3(RUN_WG), 0, grid->stream, o, b, L.dinv, L.d, L.x, L.x2, cr, run_cnt, run_base);
        } else if (mode == 3) {
            if (!run_ctl) {
                TP_HIP(hipMalloc((void **)&run_ctl, sizeof(XcdRunCtrl)));
                TP_HIP(hipMemsetAsync(run_ctl, 0, sizeof(XcdRunCtrl), grid->stream));
            }
            int P;
            const int R = xcd_rows_per_thread(L.own_n(), &P);
            if (R == 1) TP_LAUNCH((k_dia_cheb_run_xcd<DOF, 1>), dim3(8 * P), dim3(RUN_WG), 0, grid->stream, o, b, L.dinv, L.d, L.x, L.x2, cr, run_ctl, P);
            else if (R == 2) TP_LAUNCH((k_dia_cheb_run_xcd<DOF, 2>), dim3(8 * P), dim3(RUN_WG), 0, grid->stream, o, b, L.dinv, L.d, L.x, L.x2, cr, run_ctl, P);
            else if (R == 4) TP_LAUNCH((k_dia_cheb_run_xcd<DOF, 4>), dim3(8 * P), dim3(RUN_WG), 0, grid->stream, o, b, L.dinv, L.d, L.x, L.x2, cr, run_ctl, P);
            else TP_LAUNCH((k_dia_cheb_run_xcd<DOF, 8>), dim3(8 * P), dim3(RUN_WG), 0, grid->stream, o, b, L.dinv, L.d, L.x, L.x2, cr, run_ctl, P);
            if (cr.nsteps & 1) std::swap(L.x, L.x2);
        } else {
            int wgs;
            const int R = run_rows_per_thread(L.own_n(), &wgs);
            if (R == 1) TP_LAUNCH((k_dia_cheb_run<DOF, 1, false>), dim3(wgs), dim3(RUN_WG), 0, grid->stream, o, b, L.dinv, L.d, L.x, L.x2, cr, run_cnt, run_base);
            else if (R == 2) TP_LAUNCH((k_dia_cheb_run<DOF, 2, false>), dim3(wgs), dim3(RUN_WG), 0, grid->stream, o, b, L.dinv, L.d, L.x, L.x2, cr, run_cnt, run_base);
            else if (R == 4) TP_LAUNCH((k_dia_cheb_run<DOF, 4, false>), dim3(wgs), dim3(RUN_WG), 0, grid->stream, o, b, L.dinv, L.d, L.x, L.x2, cr, run_cnt, run_base);
            else TP_LAUNCH((k_dia_cheb_run<DOF, 8, false>), dim3(wgs), dim3(RUN_WG), 0, grid->stream, o, b, L.dinv, L.d, L.x, L.x2, cr, run_cnt, run_base);
            run_base += (unsigned long long)(cr.nsteps - 1) * wgs;  // no barrier after the last step
            if (cr.nsteps & 1) std::swap(L.x, L.x2);
        }
        coarse_runs++;
        const long nown = L.g.owned_nodes();
        count_launch(grid, cr.nsteps * (27.0 * DOF * DOF + 6.0 * DOF) * 8.0 * nown, cr.nsteps * 2.0 * 27 * DOF * DOF * (double)nown);
        return TP_OK;
    }

    // Chebyshev(k)-Jacobi; the iterate ping-pongs between L.x and L.x2, on exit L.x holds it
    // dot_slot >= 0: the LAST step also leaves b . x in scal[dot_slot] (this rank's part; fine tile kernel only)
    int smooth(int l, const double *b, int k, bool zero_guess, int dot_slot = -1, bool first_done = false) {
        if (replicate && l == nlv - 1) {
            // coarsest level replicated on every rank: one all-gather of the right-hand side instead of a
            // halo exchange per Chebyshev step; the result comes back with its ghost planes filled
            // (the V-cycle itself enters the replicated levels in vcycle(); this branch serves direct calls on the slab's level)
            Level<DOF> &R = lv[rix(l)];
            TP_TRY(gather_owned(lv[l], b, R, R.b, 1));
            TP_TRY(smooth(rix(l), R.b, k, zero_guess));  // (first_done never holds here: vcycle does not fuse on this path)
            Level<DOF> &L = lv[l];
            TP_HIP(hipMemcpyAsync(L.x, R.x + (long)DOF * L.g.plane() * L.g.gz0, sizeof(double) * (size_t)L.ndof(),
                                  hipMemcpyDeviceToDevice, grid->stream));
            return TP_OK;
        }
        Level<DOF> &L = lv[l];
        if (cd.factored && l == cd.level && dot_slot < 0) return coarse_direct_apply(l, b);
        double theta, delta;
        cheb_window(l, &theta, &delta);
        if (!sg_capturing && dot_slot < 0 && k >= 8 && L.kind == LV_DIA && (L.no_comm || !grid->has_comm) && smooth_graphs_on())
            return smooth_replay(l, b, k, zero_guess, first_done, theta, delta);
        const double sigma = theta / delta;
        double rho = 1.0 / sigma;
        int it = 0;
        if (zero_guess && first_done) {
            it = 1;  // x = dinv b / theta is already there (written by the restriction that produced b)
        } else if (zero_guess) {
            const long pl = (long)DOF * L.g.plane();
            TP_TRY(planes_split(l, L.x, [&](int p0, int np) -> int {
                TP_LAUNCH(k_cheb_first, dim3(grid_for(pl * np)), dim3(BLK), 0, grid->stream, L.x,
                          three_term(L) ? nullptr : L.d, b, L.dinv, 1.0 / theta, pl * p0, pl * np);
                return TP_OK;
            }));
            count_launch(grid, 32.0 * L.own_n(), 2.0 * L.own_n());
            it = 1;
        }
        if (it >= 1 && dot_slot < 0) {
            const int mode = coarse_run_mode(l, k - it);
            if (mode) return coarse_run(l, b, it, k, sigma, delta, mode);
        }
        for (; it < k; it++) {
            NodeArgs a{};
            a.x = L.x;
            a.out = L.x2;
            a.b = b;
            a.d = L.d;
            a.dinv = L.dinv;
            a.prev_zero = (zero_guess && it == 1) ? 1 : 0;
            if (it == 0) {
                a.c1 = 0.0;
                a.c2 = 1.0 / theta;
            } else {
                const double rn = 1.0 / (2.0 * sigma - rho);
                a.c1 = rn * rho;
                a.c2 = 2.0 * rn / delta;
                rho = rn;
            }
            TP_TRY(halo(l, L.x));
            if (dot_slot >= 0 && it == k - 1) {
                a.partials = grid->partials;
                a.ticket = tail_ticket(grid);
                a.red_out = grid->scal + dot_slot;
                TP_TRY(op<EPI_CHEB_DOT>(l, a));
            } else {
                TP_TRY(op<EPI_CHEB>(l, a, true));
            }
            std::swap(L.x, L.x2);
        }
        return TP_OK;
    }
    // Chebyshev window of level l.  The coarsest level is a SOLVE (the reference runs a Krylov method there,
    // LinearElasticity.cc:720-731): its window spans the whole spectrum
    void cheb_window(int l, double *theta, double *delta) const {
        const Level<DOF> &L = lv[l];
        const bool solve_level = coarsest(l) && base(l) > 0;
        const double lmin = solve_level ? L.lam_min : opt.cheb_lo * L.lam, lmax = opt.cheb_hi * L.lam;
        *theta = 0.5 * (lmax + lmin);
        *delta = 0.5 * (lmax - lmin);
    }
    // can the last post-smoothing step of a V-cycle return r . z ?  (fine tile kernel, at least one fused step)
    bool can_fuse_rz() const {
        static const int fine_v = fine_version();
        static const bool off = getenv("TP_NO_FUSE_RZ") != nullptr;
        return !off && (fine_v >= 2 || fine_v == 0) && nlv > 1 && three_term(lv[0]) && opt.nsmooth >= 1;
    }

    // every rank's owned rows of `nseg` consecutive level vectors (stride src_stride / dst_stride) -> the
    // replicated global arrays.  Rank q owns global planes q*ez + (q>0) .. (q+1)*ez.
    int gather_owned(Level<DOF> &L, const double *src, Level<DOF> &R, double *dst, int nseg, long src_stride = 0,
                     long dst_stride = 0) {
        const tp_comm &c = grid->comm;
        const long pl = (long)DOF * L.g.plane();
        const long pad = pl * (L.g.ez_own + 1);  // rank 0 owns one plane more than the others
        const int per = (int)(c.cap / pad);
        if (per < 1) return TP_ERR_ARG;
        hipStream_t s = grid->stream;
        for (int s0 = 0; s0 < nseg; s0 += per) {
            const int ns = nseg - s0 < per ? nseg - s0 : per;
            for (int q = 0; q < ns; q++)
                TP_HIP(hipMemcpyAsync(c.send_lo + (long)q * pad, src + (long)(s0 + q) * src_stride + L.own_off(),
                                      sizeof(double) * (size_t)L.own_n(), hipMemcpyDeviceToDevice, s));
            {
                CommMark cm(grid, 3, s);
                if (c.allgather(c.user, (long)ns * pad)) return TP_ERR_COMM;
            }
            for (int rk = 0; rk < grid->nranks; rk++) {
                const long p0 = (long)rk * L.g.ez_own + (rk > 0 ? 1 : 0), np = L.g.ez_own + (rk == 0 ? 1 : 0);
                for (int q = 0; q < ns; q++)
                    TP_HIP(hipMemcpyAsync(dst + (long)(s0 + q) * dst_stride + pl * p0,
                                          c.gather + (long)rk * ns * pad + (long)q * pad, sizeof(double) * (size_t)(pl * np),
                                          hipMemcpyDeviceToDevice, s));
            }
        }
        return TP_OK;
    }

    // (re)build the replicated coarsest level from the ranks' owned stencil rows
    int setup_replicated() {
        if (!replicate) return TP_OK;
        for (int l = rep0; l < nlv; l++) {
            Level<DOF> &L = lv[l], &R = lv[rix(l)];
            if (L.kind != LV_DIA) return TP_ERR_STATE;   // (only stored-stencil levels have rows to gather)
            TP_TRY(gather_owned(L, L.S, R, R.S, 27 * DOF, L.ndof(), R.ndof()));
            TP_TRY(gather_owned(L, L.dinv, R, R.dinv, 1));
            if constexpr (DOF == 3) {   // the replicated copy owns every row: its own correction of the mirrored reads
                TP_LAUNCH(k_dia_sym_fix, dim3((int)((R.g.owned_nodes() + BLK - 1) / BLK)), dim3(BLK), 0, grid->stream, R.g, R.S, (long)R.ndof());
                count_launch(grid);
            }
        }
        return TP_OK;
    }

    // PCMG multiplicative V-cycle with zero initial guesses; result in lv[l].x
    // the fine level's pre-smoothing of a V-cycle for the right-hand side `b`, enqueued AHEAD of the cycle: the Krylov
    // loop issues it for the next iteration before it waits for this iteration's residual norm, so that the device
    // has ~200 us of work while the host wakes up (vcycle(0, b) then starts behind it).  It writes multigrid scratch
    // only; if the iteration turns out to be the last one, it was for nothing.
    const double *head_for = nullptr;
    bool fine_first_done = false;  // the CG update has written x1 = dinv b / theta of the fine level already (solve())
    int vcycle_head(const double *b) {
        head_for = nullptr;
        if (nlv < 2) return TP_OK;
        TP_TRY(smooth(0, b, opt.nsmooth, true, -1, fine_first_done));
        head_for = b;
        return TP_OK;
    }
    // cycles[l]: how often level l + 1 is cycled per visit of level l (1 = V, 2 = W: PCMGSetCycleType /
    // PCMGSetCycleTypeOnLevel).  As PCMGMCycle_Private does it: the coarser level's iterate is zeroed once, further cycles
    // run on the same right-hand side from the iterate (zero_guess = false: the pre-smoother's non-zero-guess branch);
    // one cycle only into the coarsest level.
    int cycles[TP_MAX_LEVELS + 1] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
    int vcycle(int l, const double *b, int dot_slot = -1, bool first_done = false, bool zero_guess = true) {
        Level<DOF> &L = lv[l];
        if (coarsest(l)) return smooth(l, b, opt.ncoarse, zero_guess, -1, first_done);
        if (l == 0 && head_for == b) head_for = nullptr;  // pre-smoothed already (vcycle_head)
        else TP_TRY(smooth(l, b, opt.nsmooth, zero_guess, -1, first_done));
        {
            NodeArgs a{};
            a.x = L.x;
            a.out = L.r;
            a.b = b;
            TP_TRY(halo(l, L.x));
            TP_TRY(op<EPI_RESID>(l, a, true));
        }
        Level<DOF> &C = lv[l + 1];   // (slab levels and replicated slots alike: the next coarser level sits in the next slot)
        TP_TRY(halo(l, L.r));
        // from here down the levels are the replicated copies: restriction into the slab's part of the right-hand side, one
        // all-gather, the coarser levels on every rank without any exchange, the slab's window of the result back
        const bool enter_rep = replicate && !is_rep(l) && l + 1 == rep0;
        const int cyc = coarsest(l + 1) ? 1 : cycles[base(l)];
        // the restriction also takes the coarse level's first Chebyshev step from the zero guess (one launch less per
        // level and V-cycle); not when the coarse level is the replicated copy, whose right-hand side is gathered first
        static const bool no_fuse_first = getenv("TP_NO_FUSE_FIRST") != nullptr;
        const bool fuse_first = !no_fuse_first && !enter_rep && !(cd.factored && coarsest(l + 1)) &&
                                (coarsest(l + 1) ? opt.ncoarse : opt.nsmooth) >= 1;
        double th = 1.0, de = 1.0;
        if (fuse_first) cheb_window(l + 1, &th, &de);
        auto restrict_planes = [&](int p0, int np) -> int {
            const long cpl = C.g.plane();
            TP_LAUNCH((k_restrict<DOF>), dim3((int)((cpl * np + BLK - 1) / BLK)), dim3(BLK), 0, grid->stream, C.g, L.g, L.r, C.b,
                      fuse_first ? C.dinv : nullptr, fuse_first ? C.x : nullptr, fuse_first && !three_term(C) ? C.d : nullptr,
                      1.0 / th, cpl * (p0 - C.g.own_lo), cpl * np);
            return TP_OK;
        };
        if (fuse_first) TP_TRY(planes_split(l + 1, C.x, restrict_planes));  // the coarse level's first iterate is read with ghosts next
        else TP_TRY(restrict_planes(C.g.own_lo, C.g.own_hi - C.g.own_lo + 1));
        count_launch(grid, 8.0 * DOF * (L.g.owned_nodes() + C.g.owned_nodes()), 2.0 * 27 * DOF * C.g.owned_nodes());
        if (enter_rep) {
            const int r = rix(l + 1);
            Level<DOF> &R = lv[r];
            TP_TRY(gather_owned(C, C.b, R, R.b, 1));
            TP_TRY(vcycle(r, R.b, -1, false));
            for (int c = 1; c < cyc; c++) TP_TRY(vcycle(r, R.b, -1, false, false));
            TP_HIP(hipMemcpyAsync(C.x, R.x + (long)DOF * C.g.plane() * C.g.gz0, sizeof(double) * (size_t)C.ndof(),
                                  hipMemcpyDeviceToDevice, grid->stream));   // (own planes and ghosts)
        } else {
            TP_TRY(vcycle(l + 1, C.b, -1, fuse_first));
            for (int c = 1; c < cyc; c++) TP_TRY(vcycle(l + 1, C.b, -1, false, false));
            TP_TRY(halo(l + 1, C.x));
        }
        TP_TRY(planes_split(l, L.x, [&](int p0, int np) -> int {
            const long fpl = L.g.plane();
            TP_LAUNCH((k_prolong_add<DOF>), dim3((int)((fpl * np + BLK - 1) / BLK)), dim3(BLK), 0, grid->stream, C.g, L.g, C.x,
                      L.x, fpl * (p0 - L.g.own_lo), fpl * np);
            return TP_OK;
        }));
        count_launch(grid, 8.0 * DOF * (2 * L.g.owned_nodes() + C.g.owned_nodes()), 2.0 * 8 * DOF * L.g.owned_nodes());
        return smooth(l, b, opt.nsmooth, false, l == 0 ? dot_slot : -1);
    }

    // Jacobi diagonal + Chebyshev bound of a matrix-free level
    double fine_bound = 0.0;
    int setup_matfree_level(int l, const double *h_KE) {
        Level<DOF> &L = lv[l];
        MatfreeOp<DOF> o{L.KE, L.E, L.mask, L.g};
        TP_LAUNCH((k_matfree_diag<DOF>), dim3((int)((L.g.owned_nodes() + BLK - 1) / BLK)), dim3(BLK), 0,
                           grid->stream, o, L.dinv);
        count_launch(grid, 8.0 * DOF * L.g.owned_nodes() + 8.0 * L.g.own_elems(), 16.0 * DOF * L.g.owned_nodes());
        if (l == 0) {
            // the element matrix of a solver never changes: the bound (a 24 x 24 Jacobi eigenvalue iteration on the host,
            // ~90 us during which the device had nothing queued) is computed once
            if (!(fine_bound > 0.0)) {
                const double lb = elem_lambda_bound(8 * DOF, h_KE);
                fine_bound = lb > 1.0 ? lb : 1.0;
            }
            L.lam = fine_bound;
        }
        return TP_OK;
    }

    // sum over ranks of n device doubles (chunks of the 16-double framework buffer)
    int allreduce_dev(double *p, int n, bool local_only = false) {
        if (!grid->has_comm || local_only) return TP_OK;
        for (int o = 0; o < n; o += 16) {
            const int c = n - o < 16 ? n - o : 16;
            TP_HIP(hipMemcpyAsync(grid->comm.red, p + o, sizeof(double) * c, hipMemcpyDeviceToDevice, grid->stream));
            {
                CommMark cm(grid, 2, grid->stream);
                if (grid->comm.allreduce_sum(grid->comm.user, c)) return TP_ERR_COMM;
            }
            TP_HIP(hipMemcpyAsync(p + o, grid->comm.red, sizeof(double) * c, hipMemcpyDeviceToDevice, grid->stream));
        }
        return TP_OK;
    }

    // Extreme Ritz values of `steps` Lanczos iterations on D^-1/2 A D^-1/2 with FULL
    // reorthogonalisation (classical Gram-Schmidt twice): the estimates are then reproducible to
    // ~1e-13 between implementations, which the residual-history parity needs.  All coefficients
    // stay on the device; one host read at the end.
    // Enqueues a Lanczos run for level l on grid->stream (everything stays on the device, coefficients are copied to
    // the level's pinned host buffer at the end); lanczos_finish evaluates them once the stream has drained.
    int lanczos_enqueue(int l, int steps) {
        Level<DOF> &L = lv[l];
        LanBuf &B = lan[l];
        if (steps > 128) steps = 128;
        const long off = L.own_off(), n = L.own_n(), nd = L.ndof();
        // small levels: one workgroup per dot product writes its result directly (no second reduction stage)
        const int nb = n <= 65536 ? 1 : grid_for(n, 256);
        const int gn = (int)((L.g.owned_nodes() + BLK - 1) / BLK);
        hipStream_t s = grid->stream;
        const size_t need = (size_t)nd * (size_t)(steps + 1);
        if (need > B.cap) {
            (void)hipFree(B.V);
            B.V = nullptr;
            B.cap = 0;
            TP_HIP(hipMalloc((void **)&B.V, sizeof(double) * need));
            B.cap = need;
            // zeroed once: the chain reads and writes the owned range of every basis vector only
            TP_HIP(hipMemsetAsync(B.V, 0, sizeof(double) * need, s));
        }
        if (!B.coef) TP_HIP(hipMalloc((void **)&B.coef, sizeof(double) * 520));
        if (!B.part) TP_HIP(hipMalloc((void **)&B.part, sizeof(double) * 256 * 130));
        if (!B.hc) TP_HIP(hipHostMalloc((void **)&B.hc, sizeof(double) * 520));
        // Round 6: the reductions of the chain end inside the kernels that produce them (TP_LANCZOS_TAILS=0: second launches, as
        // before) -- per step 3 launches of k_reduce_multi less, |w|^2 from the second Gram-Schmidt subtraction instead of a dot
        // product of its own, and on the level-1 operator the D^-1/2 scaling in its epilogue: 12 -> 6 dependent launches per step
        // on level 1, 10 -> 6 on the stencil levels, 7 -> 6 where one workgroup per vector does the dot products.
        static const bool tails = !(getenv("TP_LANCZOS_TAILS") && atoi(getenv("TP_LANCZOS_TAILS")) == 0) && !getenv("TP_NO_REDUCE_TAIL");
        if (tails && !B.ticket) {
            TP_HIP(hipMalloc((void **)&B.ticket, sizeof(unsigned) * TICKET_WORDS));
            TP_HIP(hipMalloc((void **)&B.mticket, sizeof(unsigned) * (size_t)MT_WORDS * 130));
            TP_HIP(hipMemsetAsync(B.ticket, 0, sizeof(unsigned) * TICKET_WORDS, s));
            TP_HIP(hipMemsetAsync(B.mticket, 0, sizeof(unsigned) * (size_t)MT_WORDS * 130, s));
        }
        double *V = B.V, *coef = B.coef, *part = B.part;
        unsigned *mt = tails ? B.mticket : nullptr, *tk = tails ? B.ticket : nullptr;
        auto multi_dot = [&](const double *A, int nv, const double *wv, double *out) -> int {
            TP_LAUNCH(k_multi_dot, dim3(nb, nv), dim3(BLK), 0, s, A, nd, nv, wv, off, n, nb == 1 ? out : part, nb == 1 ? nullptr : mt, out);
            if (nb > 1 && !mt) TP_LAUNCH(k_reduce_multi, dim3(nv), dim3(BLK), 0, s, part, nb, nv, out);
            return TP_OK;
        };
        // coef: h1[129] h2[129] alpha[128] beta[128] bb[1]
        double *h1 = coef, *h2 = coef + 129, *al = coef + 258, *be = coef + 386, *bb = coef + 514;
        double *w = L.d, *t = L.r, *dis = L.b;  // scratch that smooth() never swaps: stable addresses for the graph
        TP_LAUNCH((k_lanczos_init<DOF>), dim3(gn), dim3(BLK), 0, s, L.g, V, dis, L.dinv, coef, 520);
        TP_TRY(multi_dot(V, 1, V, bb));
        TP_TRY(allreduce_dev(bb, 1, L.no_comm));
        TP_LAUNCH(k_lanczos_next, dim3(grid_for(n)), dim3(BLK), 0, s, V, bb, 0, be, V, off, n, dis, t);  // normalise v0
        // w = D^-1/2 A D^-1/2 v_j: the first scaling is written by k_lanczos_next together with v_j, the second one
        // by the operator's epilogue where the level is a stored stencil or the level-1 pattern (NodeArgs::dinv of EPI_APPLY)
        const bool scaled_apply = L.kind == LV_DIA || (DOF == 3 && L.kind == LV_MACRO && tails);
        const int ga = grid_for(n);
        for (int j = 0; j < steps; j++) {
            if (scaled_apply) {
                TP_TRY(halo(l, t));
                NodeArgs a{};
                a.x = t;
                a.out = w;
                a.dinv = dis;
                TP_TRY(op<EPI_APPLY>(l, a));
            } else {
                TP_TRY(apply(l, t, w));
                TP_LAUNCH(k_pw_mult, dim3(grid_for(n)), dim3(BLK), 0, s, w + off, dis + off, w + off, n);
            }
            for (int pass = 0; pass < 2; pass++) {
                double *h = pass ? h2 : h1;
                TP_TRY(multi_dot(V, j + 1, w, h));
                TP_TRY(allreduce_dev(h, j + 1, L.no_comm));
                // the second pass also records alpha[j] = h1[j] + h2[j] -- and, with the tails, |w|^2 of what it leaves
                if (pass && tails)
                    TP_LAUNCH(k_multi_axpy<true>, dim3(ga), dim3(BLK), 0, s, V, nd, j + 1, h, w, off, n, h1, al, part, tk, bb);
                else
                    TP_LAUNCH(k_multi_axpy<false>, dim3(ga), dim3(BLK), 0, s, V, nd, j + 1, h, w, off, n,
                              pass ? h1 : nullptr, al, nullptr, nullptr, nullptr);
            }
            if (!tails) TP_TRY(multi_dot(w, 1, w, bb));
            TP_TRY(allreduce_dev(bb, 1, L.no_comm));
            TP_LAUNCH(k_lanczos_next, dim3(grid_for(n)), dim3(BLK), 0, s, w, bb, j, be, V + (size_t)(j + 1) * nd,
                               off, n, dis, t);
            grid->launches += tails ? 6 : ((nb == 1 ? 8 : 11) - (scaled_apply ? 1 : 0));
        }
        B.m = steps;
        TP_HIP(hipMemcpyAsync(B.hc, coef, sizeof(double) * 520, hipMemcpyDeviceToHost, s));
        return TP_OK;
    }
    // The coarsest level's run as ONE launch on one XCD (coarse_run.h: k_lanczos_run_xcd): where the Chebyshev run of the
    // level qualifies for the one-XCD form; TP_NO_LANCZOS_XCD=1 keeps the chain of launches.
    XcdRunCtrl *lan_ctl = nullptr;
    bool lanczos_xcd_ok(int l, int steps) const {
        if (getenv("TP_NO_LANCZOS_XCD") || tp_xcd_disabled() || steps > LAN_MAXS || steps < 2) return false;
        if (!(l == cd_level() && base(l) > 0)) return false;
        return xcd_eligible(l, LAN_XS, 4);  // (8 rows per thread: the basis no longer fits the LDS)
    }
    int lanczos_xcd(int l, int steps) {
        Level<DOF> &L = lv[l];
        LanBuf &B = lan[l];
        hipStream_t s = grid->stream;
        if (!B.coef) TP_HIP(hipMalloc((void **)&B.coef, sizeof(double) * 520));
        if (!B.part) TP_HIP(hipMalloc((void **)&B.part, sizeof(double) * 256 * 130));
        if (!B.hc) TP_HIP(hipHostMalloc((void **)&B.hc, sizeof(double) * 520));
        if (!lan_ctl) {
            TP_HIP(hipMalloc((void **)&lan_ctl, sizeof(XcdRunCtrl)));
            TP_HIP(hipMemsetAsync(lan_ctl, 0, sizeof(XcdRunCtrl), s));
        }
        TP_HIP(hipMemsetAsync(B.coef, 0, sizeof(double) * 520, s));
        DiaOp<DOF> o{L.S, L.ndof(), L.g};
        int P;
        const int R = xcd_rows_per_thread(L.own_n(), &P);
        if (R == 1) TP_LAUNCH((k_lanczos_run_xcd<DOF, 1>), dim3(8 * P), dim3(RUN_WG), 0, s, o, L.dinv, B.part, B.coef + 258, B.coef + 386, steps, lan_ctl, P);
        else if (R == 2) TP_LAUNCH((k_lanczos_run_xcd<DOF, 2>), dim3(8 * P), dim3(RUN_WG), 0, s, o, L.dinv, B.part, B.coef + 258, B.coef + 386, steps, lan_ctl, P);
        else TP_LAUNCH((k_lanczos_run_xcd<DOF, 4>), dim3(8 * P), dim3(RUN_WG), 0, s, o, L.dinv, B.part, B.coef + 258, B.coef + 386, steps, lan_ctl, P);
        grid->launches += 1;
        B.m = steps;
        TP_HIP(hipMemcpyAsync(B.hc, B.coef, sizeof(double) * 520, hipMemcpyDeviceToHost, s));
        return TP_OK;
    }
    // is the captured chain of level l valid for the vectors it would run on now?
    bool lanczos_graph_replayable(int l) const {
        const Level<DOF> &L = lv[l];
        const void *key[5] = {L.r, L.b, L.d, L.corr, (const void *)(intptr_t)topology_epoch};
        return lan_graph_state[l] == 1 && memcmp(key, lan_graph_key[l], sizeof(key)) == 0 && !lanczos_xcd_ok(l, opt.nlanczos) &&
               getenv("TP_NO_GRAPH") == nullptr && !tp_debug_sync();
    }
    // replay (or capture, or plain enqueue) of the run of level l on grid->stream
    int lanczos_graph(int l, int steps) {
        static const bool no_graph = getenv("TP_NO_GRAPH") != nullptr || tp_debug_sync();
        Level<DOF> &L = lv[l];
        hipStream_t s = grid->stream;
        if (lanczos_xcd_ok(l, steps)) return lanczos_xcd(l, steps);
        // the chain reads/writes these vectors by address, and set_bc may rebuild the correction lists
        const void *key[5] = {L.r, L.b, L.d, L.corr, (const void *)(intptr_t)topology_epoch};
        if (lan_graph_state[l] == 1 && memcmp(key, lan_graph_key[l], sizeof(key)) != 0) {
            (void)hipGraphExecDestroy(lan_graph[l]);
            lan_graph[l] = nullptr;
            lan_graph_state[l] = 0;
        }
        if (no_graph || lan_graph_state[l] < 0) return lanczos_enqueue(l, steps);
        if (lan_graph_state[l] == 1) {
            lan[l].m = steps;
            const long launches = grid->launches;
            (void)launches;
            if (hipGraphLaunch(lan_graph[l], s) == hipSuccess) return TP_OK;
            lan_graph_state[l] = -1;
            return lanczos_enqueue(l, steps);
        }
        // first use: allocate outside the capture (a warm-up run), then capture the identical chain
        int rc = lanczos_enqueue(l, steps);
        if (rc) return rc;
        // the legacy default stream cannot capture: the chain is recorded on the spare stream and replayed where it belongs
        hipStream_t cs = (s == nullptr && side_stream) ? side_stream : s;
        if (hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            lan_graph_state[l] = -1;
            return TP_OK;  // the warm-up run above already did the work
        }
        const long l0 = grid->launches;
        const double b0 = grid->alg_bytes, f0 = grid->flops;
        grid->stream = cs;
        rc = lanczos_enqueue(l, steps);
        grid->stream = s;
        grid->launches = l0;  // the captured chain was not executed
        grid->alg_bytes = b0;
        grid->flops = f0;
        hipGraph_t g = nullptr;
        const hipError_t e1 = hipStreamEndCapture(cs, &g);
        if (rc || e1 != hipSuccess || !g || hipGraphInstantiate(&lan_graph[l], g, nullptr, nullptr, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (g) (void)hipGraphDestroy(g);
            lan_graph[l] = nullptr;
            lan_graph_state[l] = -1;
            return rc;
        }
        (void)hipGraphDestroy(g);
        memcpy(lan_graph_key[l], key, sizeof(key));
        lan_graph_state[l] = 1;
        return TP_OK;
    }
    void lanczos_finish(int l, double *lam_out, double *lam_min_out = nullptr) {
        const LanBuf &B = lan[l];
        const double *ha = B.hc + 258, *hb = B.hc + 386;
        int m = B.m;
        for (int j = 0; j < m; j++)  // breakdown (invariant subspace): truncate like the CPU path
            if (!(hb[j] > 1e-14 * fabs(ha[j]))) {
                m = j + 1;
                break;
            }
        *lam_out = tridiag_lmax(m, ha, hb);
        if (lam_min_out) *lam_min_out = tridiag_lmin(m, ha, hb);
    }
    int lanczos(int l, int steps, double *lam_out, double *lam_min_out = nullptr) {
        if (lanczos_xcd_ok(l, steps)) TP_TRY(lanczos_xcd(l, steps));
        else TP_TRY(lanczos_enqueue(l, steps));
        TP_HIP(hipStreamSynchronize(grid->stream));
        lanczos_finish(l, lam_out, lam_min_out);
        return TP_OK;
    }

    // z = M r : one V-cycle.  Returns the pointer holding z (lv[0].x).
    int precond(const double *r, double **z, int dot_slot = -1) {
        if (opt.ksp_mode == 1) return refksp_precond(*this, r, z);
        TP_TRY(vcycle(0, r, dot_slot));
        TP_TRY(drain_halos());
        *z = lv[0].x;
        return TP_OK;
    }

    // KSPSolve, KSPCG with the unpreconditioned norm, reference norm ||b||
    // (KSPConvergedDefault with a nonzero initial guess)
    int solve(const double *b, double *x, int *its_out, double *rnorm_out, double *bnorm_out, double *hist,
              int hist_cap) {
        if (!ready) return TP_ERR_STATE;
        if (opt.ksp_mode == 1) return refksp_solve(*this, b, x, its_out, rnorm_out, bnorm_out, hist, hist_cap);
        Level<DOF> &L = lv[0];
        hipStream_t s = grid->stream;
        const long off = L.own_off(), n = L.own_n();
        const int nb = grid_for(n, 2048);  // one resident round of workgroups; the reduction tail wants few arrivals
        double *r = cg_r, *p = cg_p, *p_alt = cg_p2, *w = cg_w;
        {
            NodeArgs a{};
            a.x = x;
            a.out = r;
            a.b = b;
            TP_TRY(halo(0, x));
            if (SYMKE_KRYLOV && DOF == 3 && L.use_tile) {
                // the initial residual belongs to the Krylov method: its operator is the Krylov product (matfree_tile.h:
                // SYMKE_KRYLOV -- KE's action on the iterate's translation part included), not the V-cycle's residual kernel
                a.b = nullptr;
                a.partials = grid->partials;
                a.ticket = tail_ticket(grid);
                a.red_out = grid->scal + S_PW;   // (x0 . A x0: not used)
                TP_TRY(op<EPI_APPLY_DOT>(0, a));
                TP_TRY(finish_tail<1>(grid, last_nblocks, S_PW));
                TP_LAUNCH(k_axpby, dim3(grid_for(n)), dim3(BLK), 0, s, r + off, 1.0, b + off, -1.0, n);   // r = b - A x
                count_launch(grid, 24.0 * n, 1.0 * n);
            } else {
                TP_TRY(op<EPI_RESID>(0, a));
            }
        }
        TP_LAUNCH(k_dot2, dim3(nb), dim3(BLK), 0, s, b, b, r, r, off, n, grid->partials);
        count_launch(grid, 16.0 * n, 4.0 * n);
        TP_TRY(reduce_partials<2>(grid, nb, S_BB));
        double v2[2];
        // the first V-cycle's fine pre-smoothing goes out BEFORE the host waits for the two norms (as the loop does for every
        // later iteration): the device has work while the host wakes up; wasted only when the warm start is converged already
        static const bool spec_head0 = getenv("TP_NO_SPEC_HEAD") == nullptr;
        head_for = nullptr;
        fine_first_done = false;  // (a previous solve that returned through TP_TRY inside its loop may have left it set)
        // on slabs a wasted head also wastes its halo exchanges: not when max_it = 0 or the previous solve needed no iteration
        if (spec_head0 && opt.ksp_mode == 0 && nlv >= 2 && !sg_capturing && opt.max_it > 0 && !(grid->has_comm && last_solve_its == 0)) {
            TP_TRY(read_scal_begin(grid, S_BB, 2));
            TP_TRY(vcycle_head(r));
            TP_TRY(read_scal_end(grid, 2, v2));
        } else {
            TP_TRY(read_scal(grid, S_BB, 2, v2));
        }
        const double bnorm = sqrt(v2[0]);
        double rnorm = sqrt(v2[1]);
        const double ttol = fmax(opt.rtol * bnorm, opt.atol);
        if (bnorm_out) *bnorm_out = bnorm;
        if (hist && hist_cap > 0) hist[0] = rnorm;
        int its = 0, rc = TP_OK;
        int rz_cur = S_RZ0, rz_old = S_RZ1;
        static const bool spec_head = getenv("TP_NO_SPEC_HEAD") == nullptr;
        static const bool fuse_cg = getenv("TP_NO_CG_FUSE") == nullptr;
        if (rnorm > ttol) {
            for (its = 1; its <= opt.max_it; its++) {
                double *z;
                if (can_fuse_rz()) {  // r . z comes out of the V-cycle's last smoothing step
                    TP_TRY(precond(r, &z, rz_cur));
                    TP_TRY(finish_tail<1>(grid, last_nblocks, rz_cur));
                } else {
                    TP_TRY(precond(r, &z));
                    TP_TRY(dot_to_slot(grid, r + off, z + off, n, rz_cur));
                }
                // one rank, second-generation fine kernel: p = z + beta p inside the product's launch (fine_tile.h: the staged
                // input is fma(beta, p_old, z), the owner of a node stores it to the other p buffer) -- one pass over p and z
                // and one launch less per iteration; same values, bit for bit
                const bool fuse_p = fuse_cg && !grid->has_comm && runs_fine_tile(L) && !sg_capturing;
                if (fuse_p) {
                    NodeArgs a{};
                    a.x = its == 1 ? z : p;   // (first iteration: p = z -- beta 0 on z itself, whatever the buffer holds)
                    a.pz = z;
                    a.pnew = p_alt;
                    a.pscal = its == 1 ? nullptr : grid->scal;
                    a.slot_new = rz_cur;
                    a.slot_old = rz_old;
                    a.out = w;
                    a.partials = grid->partials;
                    a.ticket = tail_ticket(grid);
                    a.red_out = grid->scal + S_PW;
                    TP_TRY(op<EPI_APPLY_DOT>(0, a));
                    TP_TRY(finish_tail<1>(grid, last_nblocks, S_PW));
                    std::swap(p, p_alt);
                } else {
                    {
                        const long pl = (long)DOF * L.g.plane();
                        TP_TRY(planes_split(0, p, [&](int p0, int np) -> int {
                            TP_LAUNCH(k_cg_update_p, dim3(grid_for(pl * np)), dim3(BLK), 0, s, p, z, grid->scal, rz_cur, rz_old,
                                      its == 1 ? 1 : 0, pl * p0, pl * np);
                            return TP_OK;
                        }));
                    }
                    count_launch(grid, 24.0 * n, 2.0 * n);
                    NodeArgs a{};
                    a.x = p;
                    a.out = w;
                    a.partials = grid->partials;
                    a.ticket = tail_ticket(grid);
                    a.red_out = grid->scal + S_PW;
                    TP_TRY(halo(0, p));
                    TP_TRY(op<EPI_APPLY_DOT>(0, a));
                    TP_TRY(finish_tail<1>(grid, last_nblocks, S_PW));
                }
                // one rank: ||r||^2 goes straight to pinned host memory from the reduction's last workgroup (no copy in
                // the stream), and the update also writes the first Chebyshev step of the next V-cycle (see the kernel)
                const bool direct_rr = fuse_cg && !grid->has_comm && tail_ticket(grid) && grid->h_scal_dev;
                const bool fuse_first = direct_rr && spec_head && its < opt.max_it && nlv >= 2 && opt.nsmooth >= 1 && three_term(L) &&
                                        !sg_capturing;
                double th0 = 1.0, de0 = 1.0;
                if (fuse_first) cheb_window(0, &th0, &de0);
                static const bool cg_nt = !(getenv("TP_CG_NT") != nullptr && atoi(getenv("TP_CG_NT")) == 0);  // on (TP_CG_NT=0: plain loads / stores)
                if (cg_nt && n >= (1L << 22))   // (below ~32 MB a vector lives in the caches anyway: the hint costs 0.5 % at C1 / C2)
                    TP_LAUNCH(k_cg_update_xr<true>, dim3(nb), dim3(BLK), 0, s, x, r, p, w, grid->scal, rz_cur, off, n,
                              grid->partials, tail_ticket(grid), grid->scal + S_RR, direct_rr ? grid->h_scal_dev : nullptr,
                              fuse_first ? L.x : nullptr, L.dinv, 1.0 / th0);
                else
                    TP_LAUNCH(k_cg_update_xr<false>, dim3(nb), dim3(BLK), 0, s, x, r, p, w, grid->scal, rz_cur, off, n,
                              grid->partials, tail_ticket(grid), grid->scal + S_RR, direct_rr ? grid->h_scal_dev : nullptr,
                              fuse_first ? L.x : nullptr, L.dinv, 1.0 / th0);
                count_launch(grid, (fuse_first ? 64.0 : 48.0) * n, 6.0 * n);
                TP_TRY(finish_tail<1>(grid, nb, S_RR));
                double rr;
                if (direct_rr) {
                    if (!grid->ev_scal) TP_HIP(hipEventCreateWithFlags(&grid->ev_scal, hipEventDisableTiming));
                    TP_HIP(hipEventRecord(grid->ev_scal, s));
                } else {
                    TP_TRY(read_scal_begin(grid, S_RR, 1));
                }
                fine_first_done = fuse_first;
                if (spec_head && its < opt.max_it) TP_TRY(vcycle_head(r));  // next iteration's first kernels, then wait
                fine_first_done = false;
                TP_TRY(read_scal_end(grid, 1, &rr));
                rnorm = sqrt(rr);
                if (hist && its < hist_cap) hist[its] = rnorm;
                if (rnorm <= ttol) break;
                if (!(rnorm <= opt.dtol * bnorm)) {  // also catches NaN
                    rc = TP_ERR_DIVERGED;
                    break;
                }
                if (its == opt.max_it) break;
                std::swap(rz_cur, rz_old);
            }
        }
        head_for = nullptr;
        TP_TRY(drain_halos());
        gaveup_seen = false;
        if (rc == TP_ERR_DIVERGED) gaveup_seen = xcd_gaveup();  // (before the control blocks are cleared below)
        if (rc == TP_ERR_DIVERGED && run_cnt) {
            // a multi-workgroup coarse run that gave up leaves its give-up flag set and fewer arrivals than run_base
            // assumes: every later run would time out as well.  Start the counters over.
            TP_HIP(hipMemsetAsync(run_cnt, 0, 2 * sizeof(unsigned long long), s));
            run_base = 0;
        }
        if (rc == TP_ERR_DIVERGED && run_ctl) TP_HIP(hipMemsetAsync(run_ctl, 0, sizeof(XcdRunCtrl), s));
        if (rc == TP_ERR_DIVERGED && lan_ctl) TP_HIP(hipMemsetAsync(lan_ctl, 0, sizeof(XcdRunCtrl), s));
        if (rc == TP_ERR_DIVERGED && cd.ctl) TP_HIP(hipMemsetAsync(cd.ctl, 0, sizeof(XcdRunCtrl), s));
        if (its_out) *its_out = its;
        if (rnorm_out) *rnorm_out = rnorm;
        last_solve_its = its;
        return rc;
    }
    int last_solve_its = -1;  // iteration count of the previous solve (-1: none yet)
};

// operators.h -- level operators and the node-centric kernels built on them.
//
// Two operator kinds (both applied by GATHER, one thread per node, no atomics,
// coalesced stores, bitwise reproducible):
//   MatfreeOp<DOF>: y_i = sum over the <=8 elements around node i of
//                   E_e * KE[rows of i] * (N u)_e, then the Dirichlet rows
//                   y = N y + (I-N) u           (LinearElasticity.cc:510-542)
//   DiaOp<DOF>    : 27-point block stencil stored by diagonals (Galerkin levels)
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------
template <int DOF>
struct MatfreeOp {
    const double *__restrict__ KE;      // (8*DOF)^2, row major, wave-uniform -> scalar loads
    const double *__restrict__ E;       // per stored element scale, or nullptr (== 1)
    const uint8_t *__restrict__ mask;   // per node: bit c set = dof c clamped; or nullptr
    Geom g;

    __device__ inline void load_masked(const double *__restrict__ u, long nb, double ub[DOF]) const {
#pragma unroll
        for (int c = 0; c < DOF; c++) ub[c] = u[nb * DOF + c];
        if (mask) {
            const unsigned m = mask[nb];
#pragma unroll
            for (int c = 0; c < DOF; c++)
                if ((m >> c) & 1u) ub[c] = 0.0;
        }
    }

    __device__ inline void apply(const double *__restrict__ u, int i, int j, int k, long n, double y[DOF]) const {
        constexpr int ED = 8 * DOF;
#pragma unroll
        for (int r = 0; r < DOF; r++) y[r] = 0.0;
        // this node is local corner `a` of the element at (i-LX[a], j-LY[a], k-LZ[a])
#pragma unroll
        for (int a = 0; a < 8; a++) {
            const int ei = i - LXc(a), ej = j - LYc(a), ek = k - LZc(a);
            if (ei < 0 || ei >= g.ex || ej < 0 || ej >= g.ey || ek < 0 || ek >= g.ezl) continue;
            const double Ee = E ? E[(long)ei + (long)g.ex * (ej + (long)g.ey * ek)] : 1.0;
            double s[DOF];
#pragma unroll
            for (int r = 0; r < DOF; r++) s[r] = 0.0;
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const long nb = (long)(ei + LXc(b)) + (long)g.nx * ((ej + LYc(b)) + (long)g.ny * (ek + LZc(b)));
                double ub[DOF];
                load_masked(u, nb, ub);
#pragma unroll
                for (int r = 0; r < DOF; r++)
#pragma unroll
                    for (int c = 0; c < DOF; c++) s[r] = fma(KE[(a * DOF + r) * ED + b * DOF + c], ub[c], s[r]);
            }
#pragma unroll
            for (int r = 0; r < DOF; r++) y[r] = fma(Ee, s[r], y[r]);
        }
        if (mask) {
            const unsigned m = mask[n];
#pragma unroll
            for (int r = 0; r < DOF; r++)
                if ((m >> r) & 1u) y[r] = u[n * DOF + r];
        }
    }
    // the node's own DOF x DOF block of N K N + (I - N), row major (Gauss-Seidel sweeps, refksp.h)
    __device__ inline void diag_block(long n, int i, int j, int k, double D[DOF * DOF]) const {
        constexpr int ED = 8 * DOF;
#pragma unroll
        for (int q = 0; q < DOF * DOF; q++) D[q] = 0.0;
#pragma unroll
        for (int a = 0; a < 8; a++) {
            const int ei = i - LXc(a), ej = j - LYc(a), ek = k - LZc(a);
            if (ei < 0 || ei >= g.ex || ej < 0 || ej >= g.ey || ek < 0 || ek >= g.ezl) continue;
            const double Ee = E ? E[(long)ei + (long)g.ex * (ej + (long)g.ey * ek)] : 1.0;
#pragma unroll
            for (int r = 0; r < DOF; r++)
#pragma unroll
                for (int c = 0; c < DOF; c++) D[r * DOF + c] = fma(Ee, KE[(a * DOF + r) * ED + a * DOF + c], D[r * DOF + c]);
        }
        if (mask) {
            const unsigned m = mask[n];
#pragma unroll
            for (int r = 0; r < DOF; r++)
#pragma unroll
                for (int c = 0; c < DOF; c++) {
                    const bool fr = (m >> r) & 1u, fc = (m >> c) & 1u;
                    if (fr || fc) D[r * DOF + c] = (r == c) ? 1.0 : 0.0;
                }
        }
    }
};

// ---------------------------------------------------------------------------
// A CONSTANT-COEFFICIENT scalar operator (the Helmholtz filter's K_f, PDEFilter.cc:251-264: one 8 x 8 element matrix for
// the whole level, no moduli, no Dirichlet rows) as the 27-point stencil it is (round 5): a node's row depends only on
// which of its 8 adjacent elements exist -- per axis "no lower element" / both / "no upper element", 27 classes -- so the
// 27 x 27 weights are tabulated once per level on the host (pde_stencil_table) and a thread does 27 loads and 27 fma
// instead of the gather form's 64 + 64 (k_node<1, MatfreeOp<1>>: 11.4 us per launch at 0.8 M nodes, 14 % of config 4's
// busy time).  Same operator, other summation order: equal to the gather form to rounding.
struct ScalarStencilOp {
    const double *__restrict__ W;   // [27 classes][27 offsets], class = (cz*3 + cy)*3 + cx, offset = (dz+1)*9 + (dy+1)*3 + (dx+1)
    Geom g;
    __device__ inline void apply(const double *__restrict__ u, int i, int j, int k, long n, double y[1]) const {
        // element (i-1 .. i) x (j-1 .. j) x (k-1 .. k) existence, as MatfreeOp tests it (local layers [0, ezl))
        const int cx = i == 0 ? 0 : (i >= g.ex ? 2 : 1), cy = j == 0 ? 0 : (j >= g.ey ? 2 : 1), cz = k == 0 ? 0 : (k >= g.ezl ? 2 : 1);
        const double *__restrict__ w = W + ((cz * 3 + cy) * 3 + cx) * 27;
        double s = 0.0;
#pragma unroll
        for (int dz = -1; dz <= 1; dz++)
#pragma unroll
            for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                for (int dx = -1; dx <= 1; dx++) {
                    const bool ok = !((dx < 0 && cx == 0) || (dx > 0 && cx == 2) || (dy < 0 && cy == 0) || (dy > 0 && cy == 2) ||
                                      (dz < 0 && cz == 0) || (dz > 0 && cz == 2));
                    const long nb = ok ? n + dx + (long)g.nx * (dy + (long)g.ny * dz) : n;   // (weight 0 there)
                    s = fma(w[(dz + 1) * 9 + (dy + 1) * 3 + (dx + 1)], u[nb], s);
                }
        y[0] = s;
    }
};
// the table of ScalarStencilOp from an 8 x 8 element matrix (reference corner order): W[class][offset] = sum over the
// existing elements around the node, in corner order a = 0 .. 7, of KF[a][b] for the neighbour b of that element at the offset
inline void pde_stencil_table(const double *KF, double *W /* 27 * 27 */) {
    for (int q = 0; q < 27 * 27; q++) W[q] = 0.0;
    for (int cz = 0; cz < 3; cz++)
        for (int cy = 0; cy < 3; cy++)
            for (int cx = 0; cx < 3; cx++) {
                double *w = W + ((cz * 3 + cy) * 3 + cx) * 27;
                for (int a = 0; a < 8; a++) {   // the node is corner a of the element at (i - LX[a], j - LY[a], k - LZ[a])
                    const int lx = h_LX[a], ly = h_LY[a], lz = h_LZ[a];
                    // that element is the LOWER one along an axis if l = 1 (needs class != 0), the upper one if l = 0 (class != 2)
                    if ((lx ? cx == 0 : cx == 2) || (ly ? cy == 0 : cy == 2) || (lz ? cz == 0 : cz == 2)) continue;
                    for (int b = 0; b < 8; b++) {
                        const int dx = h_LX[b] - lx, dy = h_LY[b] - ly, dz = h_LZ[b] - lz;
                        w[(dz + 1) * 9 + (dy + 1) * 3 + (dx + 1)] += KF[a * 8 + b];
                    }
                }
            }
}

// ---------------------------------------------------------------------------
// Block 27-point stencil by diagonals: S[(blk*DOF + c) * nrows + row],
// row = node*DOF + r, blk = (dk+1)*9 + (dj+1)*3 + (di+1), column dof c.
template <int DOF>
struct DiaOp {
    const double *__restrict__ S;
    long nrows;  // DOF * local nodes
    Geom g;
    // rows of this launch, counted from the first owned row: [t0, t0 + tn) followed by [t1, t1 + tn1); tn < 0: all owned rows.
    // Two ranges: the launch that produces the two boundary planes of a slab first (halo overlap on the stencil levels, mg.h)
    long t0 = 0, tn = -1, t1 = 0, tn1 = 0;
    __device__ inline long rows_here() const { return tn < 0 ? g.owned_nodes() * DOF : tn + tn1; }
    __device__ inline long row_of(long tl) const { return tn < 0 ? tl : (tl < tn ? t0 + tl : t1 + (tl - tn)); }

    __device__ inline void apply(const double *__restrict__ u, int i, int j, int k, long n, double y[DOF]) const {
#pragma unroll
        for (int r = 0; r < DOF; r++) y[r] = 0.0;
#pragma unroll
        for (int dk = -1; dk <= 1; dk++) {
            if (k + dk < 0 || k + dk >= g.nzl) continue;
#pragma unroll
            for (int dj = -1; dj <= 1; dj++) {
                if (j + dj < 0 || j + dj >= g.ny) continue;
#pragma unroll
                for (int di = -1; di <= 1; di++) {
                    if (i + di < 0 || i + di >= g.nx) continue;
                    const int blk = (dk + 1) * 9 + (dj + 1) * 3 + (di + 1);
                    const long nb = n + di + (long)g.nx * (dj + (long)g.ny * dk);
#pragma unroll
                    for (int c = 0; c < DOF; c++) {
                        const double uc = u[nb * DOF + c];
                        const double *__restrict__ Sd = S + (long)(blk * DOF + c) * nrows + n * DOF;
#pragma unroll
                        for (int r = 0; r < DOF; r++) y[r] = fma(Sd[r], uc, y[r]);
                    }
                }
            }
        }
    }
    __device__ inline void diag_block(long n, int, int, int, double D[DOF * DOF]) const {
#pragma unroll
        for (int c = 0; c < DOF; c++)
#pragma unroll
            for (int r = 0; r < DOF; r++) D[r * DOF + c] = S[(long)(13 * DOF + c) * nrows + n * DOF + r];
    }
};

// ---------------------------------------------------------------------------
// epilogues of the node kernel
// EPI_CHEB_DOT (fine tile kernel only): Chebyshev step that also returns b . x_out -- the r.z of CG when it is the
// last smoothing step of the V-cycle
enum { EPI_APPLY = 0, EPI_RESID = 1, EPI_CHEB = 2, EPI_APPLY_DOT = 3, EPI_CHEB_DOT = 4 };

struct NodeArgs {
    const double *x;     // operator input
    double *out;         // APPLY: y | RESID: r | CHEB: x_out | APPLY_DOT: w
    const double *b;     // RESID, CHEB
    double *d;           // CHEB direction (in/out)
    const double *dinv;  // CHEB Jacobi | APPLY on stored-stencil levels: optional pointwise scale of y (Lanczos)
    double c1, c2;       // CHEB recurrence coefficients
    int prev_zero;       // CHEB, 3-term form: the previous iterate is the zero guess (not read)
    double *partials;    // APPLY_DOT / CHEB_DOT: per-block partials of x . (A x) / b . x_out
    unsigned *ticket;    // ... arrival counter and result of the in-kernel reduction tail (common.h)
    double *red_out;
    // APPLY_DOT of the fine tile kernel (fine_tile.h), one rank: the CG direction update fused into the product.  The staged
    // input is fma(beta, x, pz) with beta = pscal[slot_new] / pscal[slot_old] (0 if pscal is NULL), and the owner of a node
    // stores that value to pnew: p_new = z + beta p_old and w = A p_new, p . w in ONE launch.  pz = NULL: plain product.
    const double *pz;
    double *pnew;
    const double *pscal;
    int slot_new, slot_old;
};

template <int DOF, class Op, int EPI>
__global__ __launch_bounds__(BLK) void k_node(Op op, NodeArgs a) {
    const Geom &g = op.g;
    const long plane = g.plane();
    const long t = blockIdx.x * (long)BLK + threadIdx.x;
    double pdot = 0.0;
    if (t < g.owned_nodes()) {
        const int k = g.own_lo + (int)(t / plane);
        const int rem = (int)(t % plane);
        const int j = rem / g.nx, i = rem % g.nx;
        const long n = t + plane * g.own_lo;
        double y[DOF];
        op.apply(a.x, i, j, k, n, y);
#pragma unroll
        for (int r = 0; r < DOF; r++) {
            const long q = n * DOF + r;
            if (EPI == EPI_APPLY) {
                a.out[q] = y[r];
            } else if (EPI == EPI_RESID) {
                a.out[q] = a.b[q] - y[r];
            } else if (EPI == EPI_CHEB) {
                const double res = a.b[q] - y[r];
                const double dn = a.c1 * a.d[q] + a.c2 * (a.dinv[q] * res);
                a.d[q] = dn;
                a.out[q] = a.x[q] + dn;
            } else {
                a.out[q] = y[r];
                pdot = fma(a.x[q], y[r], pdot);
            }
        }
    }
    if (EPI == EPI_APPLY_DOT) {
        const double v[1] = {block_sum(pdot)};
        reduce_tail<1>(v, a.partials, gridDim.x, blockIdx.x, a.ticket, a.red_out);
    }
}

// first Chebyshev step with a zero initial guess: x = dinv*b/theta, d = x (owned range; d = null in the 3-term
// form, whose next step is told that the previous iterate is zero)
__global__ __launch_bounds__(BLK) void k_cheb_first(double *__restrict__ x, double *__restrict__ d,
                                                    const double *__restrict__ b, const double *__restrict__ dinv,
                                                    double inv_theta, long off, long n) {
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) {
        const double v = dinv[off + i] * b[off + i] * inv_theta;
        if (d) d[off + i] = v;
        x[off + i] = v;
    }
}

// Jacobi diagonal of the matrix-free operator: d_i = sum_e E_e n_i KE[ii] + (1 - n_i)
template <int DOF>
__global__ __launch_bounds__(BLK) void k_matfree_diag(MatfreeOp<DOF> op, double *__restrict__ dinv) {
    const Geom &g = op.g;
    constexpr int ED = 8 * DOF;
    const long plane = g.plane();
    const long t = blockIdx.x * (long)BLK + threadIdx.x;
    if (t >= g.owned_nodes()) return;
    const int k = g.own_lo + (int)(t / plane);
    const int rem = (int)(t % plane);
    const int j = rem / g.nx, i = rem % g.nx;
    const long n = t + plane * g.own_lo;
    double dg[DOF];
#pragma unroll
    for (int r = 0; r < DOF; r++) dg[r] = 0.0;
#pragma unroll
    for (int a = 0; a < 8; a++) {
        const int ei = i - LXc(a), ej = j - LYc(a), ek = k - LZc(a);
        if (ei < 0 || ei >= g.ex || ej < 0 || ej >= g.ey || ek < 0 || ek >= g.ezl) continue;
        const double Ee = op.E ? op.E[(long)ei + (long)g.ex * (ej + (long)g.ey * ek)] : 1.0;
#pragma unroll
        for (int r = 0; r < DOF; r++) dg[r] = fma(Ee, op.KE[(a * DOF + r) * ED + a * DOF + r], dg[r]);
    }
    const unsigned m = op.mask ? op.mask[n] : 0u;
#pragma unroll
    for (int r = 0; r < DOF; r++) dinv[n * DOF + r] = ((m >> r) & 1u) ? 1.0 : 1.0 / dg[r];
}

// ---------------------------------------------------------------------------
// grid transfer: trilinear Q1, coarse node I at fine node 2I
// (DMCreateInterpolation on a DMDA, LinearElasticity.cc:704); restriction = P^T.
// ---------------------------------------------------------------------------
// coarse b_c[I] = sum over the 27 fine neighbours of 2I of w * r_f ; owned coarse nodes
// first != NULL: also the first Chebyshev step of the coarse level from a zero guess (k_cheb_first fused in):
// x_c = dinv_c * b_c * inv_theta (and d_c = x_c where the level keeps a direction vector)
#ifdef TP_RESTRICT_NARROW
constexpr bool RESTRICT_NARROW = true;   // A/B: round 1-5's nine 8-byte loads per row
#else
constexpr bool RESTRICT_NARROW = false;
#endif
template <int DOF>
__global__ __launch_bounds__(BLK) void k_restrict(Geom gc, Geom gf, const double *__restrict__ rf,
                                                  double *__restrict__ bc, const double *__restrict__ dinv_c = nullptr,
                                                  double *__restrict__ x_c = nullptr, double *__restrict__ d_c = nullptr,
                                                  double inv_theta = 0.0, long t0 = 0, long tn = -1) {
    // owned coarse nodes [t0, t0 + tn) counted from the first owned plane (tn < 0: all)
    const long plane = gc.plane();
    // workgroups are dealt round-robin to the 8 XCDs: give every XCD a contiguous run of coarse nodes, so that the fine
    // planes two neighbouring coarse planes share are fetched into ONE L2
    const int nbk = gridDim.x, x8 = blockIdx.x & 7;
    const long bid = (long)x8 * (nbk >> 3) + min(x8, nbk & 7) + (blockIdx.x >> 3);
    const long t = t0 + bid * (long)BLK + threadIdx.x;
    if (t >= (tn < 0 ? gc.owned_nodes() : t0 + tn)) return;
    const int K = gc.own_lo + (int)(t / plane);
    const int rem = (int)(t % plane);
    const int J = rem / gc.nx, I = rem % gc.nx;
    const long nc = t + plane * gc.own_lo;
    double s[DOF];
#pragma unroll
    for (int r = 0; r < DOF; r++) s[r] = 0.0;
    if (DOF == 3 && !RESTRICT_NARROW) {
        // Round 6: the three x-neighbours of a row are 9 CONTIGUOUS doubles per thread: four 16-byte loads and one 8-byte load
        // (8-byte aligned: legal on this target) instead of nine 8-byte loads at a lane stride of 48 bytes -- the kernel was
        // bound by vector-memory issue (63 % of the wave time issue-stalled, every load instruction touching 24-48 lines:
        // profiles/r06_pmc_counters_other_kernels.json).  Branch-free at the x-boundaries too: the window starts at
        // clamp(2I - 1, 0, nx - 3) and each of its three nodes takes the weight of its distance from 2I (0 beyond the stencil),
        // so the non-zero terms enter the sums in the order of the narrow form: the same bits.
        typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
        const int s0 = min(max(2 * I - 1, 0), gf.nx - 3);
        double wx[3];
#pragma unroll
        for (int t3 = 0; t3 < 3; t3++) {
            const int di = s0 + t3 - 2 * I;
            wx[t3] = di == 0 ? 1.0 : ((di == 1 || di == -1) ? 0.5 : 0.0);
        }
#pragma unroll
        for (int dk = -1; dk <= 1; dk++) {
            const int k = 2 * K + dk;
            const bool okk = k >= 0 && k < gf.nzl;
#pragma unroll
            for (int dj = -1; dj <= 1; dj++) {
                const int j = 2 * J + dj;
                const bool ok = okk && j >= 0 && j < gf.ny;
                const double wyz = ok ? (dj ? 0.5 : 1.0) * (dk ? 0.5 : 1.0) : 0.0;
                const long row = ok ? (long)gf.nx * (j + (long)gf.ny * k) : (long)gf.nx * (2 * J + (long)gf.ny * (2 * K));
                const double *__restrict__ q = rf + (row + s0) * 3;
                const d2u a = *(const d2u *)(q), b2 = *(const d2u *)(q + 2), c2 = *(const d2u *)(q + 4), d2 = *(const d2u *)(q + 6);
                const double v[9] = {a.x, a.y, b2.x, b2.y, c2.x, c2.y, d2.x, d2.y, q[8]};
#pragma unroll
                for (int t3 = 0; t3 < 3; t3++) {
                    // (the narrow form's weight is (x * y) * z: powers of two, exact in any order)
                    const double w = wx[t3] * wyz;
#pragma unroll
                    for (int r = 0; r < 3; r++) s[r] = fma(w, v[3 * t3 + r], s[r]);
                }
            }
        }
    } else
#pragma unroll
    for (int dk = -1; dk <= 1; dk++) {
        const int k = 2 * K + dk;
        const bool okk = k >= 0 && k < gf.nzl;
#pragma unroll
        for (int dj = -1; dj <= 1; dj++) {
            const int j = 2 * J + dj;
            const bool okj = okk && j >= 0 && j < gf.ny;
#pragma unroll
            for (int di = -1; di <= 1; di++) {
                const int i = 2 * I + di;
                const bool ok = okj && i >= 0 && i < gf.nx;
                const double w = ok ? (di ? 0.5 : 1.0) * (dj ? 0.5 : 1.0) * (dk ? 0.5 : 1.0) : 0.0;
                const long nf = ok ? (long)i + (long)gf.nx * (j + (long)gf.ny * k) : (long)(2 * I) + (long)gf.nx * (2 * J + (long)gf.ny * (2 * K));
#pragma unroll
                for (int r = 0; r < DOF; r++) s[r] = fma(w, rf[nf * DOF + r], s[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < DOF; r++) {
        bc[nc * DOF + r] = s[r];
        if (x_c) {
            const double v = dinv_c[nc * DOF + r] * s[r] * inv_theta;
            x_c[nc * DOF + r] = v;
            if (d_c) d_c[nc * DOF + r] = v;
        }
    }
}

// (Round 5, measured and dropped: the same restriction LDS tiled -- a workgroup of 16 x 4 x 4 coarse nodes stages its
// 33 x 9 x 9 fine neighbourhood with row-contiguous loads, all issued before the first LDS store, and sums out of LDS in this
// kernel's order, bit-equal.  At 128^3: 42 us against this kernel's 27.8 us (56 us with the rows staged one round trip at a
// time).  The gather's 3.4-fold re-reads are L1/L2 hits that overlap with each other; the tiled form pays a load phase, a
// barrier and a compute phase in sequence on two workgroups per CU.)
// fine x_f += P x_c ; owned fine nodes
template <int DOF>
__global__ __launch_bounds__(BLK) void k_prolong_add(Geom gc, Geom gf, const double *__restrict__ xc,
                                                     double *__restrict__ xf, long t0 = 0, long tn = -1) {
    // owned fine nodes [t0, t0 + tn) counted from the first owned plane (tn < 0: all)
    const long plane = gf.plane();
    const long t = t0 + blockIdx.x * (long)BLK + threadIdx.x;
    if (t >= (tn < 0 ? gf.owned_nodes() : t0 + tn)) return;
    const int k = gf.own_lo + (int)(t / plane);
    const int rem = (int)(t % plane);
    const int j = rem / gf.nx, i = rem % gf.nx;
    const long nf = t + plane * gf.own_lo;
    const int I0 = i >> 1, J0 = j >> 1, K0 = k >> 1;
    const int mi = i & 1, mj = j & 1, mk = k & 1;
    double s[DOF];
#pragma unroll
    for (int r = 0; r < DOF; r++) s[r] = 0.0;
    // (Round 6, measured and dropped: the same gather branch-free -- always 8 coarse triples, the absent ones with weight 0, the
    // fine triple requested with them: bit-equal, 34.1 us against 30.6 in the step for the 1 -> 0 launch: the launch is bound by
    // its read-modify-write of the fine vector out of a cold cache, not by the latency of its gathers.  Nor by the strided
    // triples: a thread per DOUBLE of a fine plane -- contiguous 512-byte runs per wave, three times the threads and gather
    // instructions -- measured +0.25 ms per design iteration, ~50 us per launch.)
    for (int kk = 0; kk <= mk; kk++)
        for (int jj = 0; jj <= mj; jj++)
            for (int ii = 0; ii <= mi; ii++) {
                const double w = (mi ? 0.5 : 1.0) * (mj ? 0.5 : 1.0) * (mk ? 0.5 : 1.0);
                const long nc = (long)(I0 + ii) + (long)gc.nx * ((J0 + jj) + (long)gc.ny * (K0 + kk));
#pragma unroll
                for (int r = 0; r < DOF; r++) s[r] = fma(w, xc[nc * DOF + r], s[r]);
            }
#pragma unroll
    for (int r = 0; r < DOF; r++) xf[nf * DOF + r] += s[r];
}

// ---------------------------------------------------------------------------
// Stencil levels, one thread per matrix ROW (node x dof): the coarse grids are
// small (17^3 .. 65^3 nodes), so the finer decomposition is what fills the chip.
// ---------------------------------------------------------------------------
// Chebyshev direction update of the stored-stencil kernels, d+ = c1 d + c2 dinv (b - y), in ONE spelled-out operation
// order: the compiler's own contraction of `c1 * d + c2 * t` differs from kernel to kernel, and the row kernels
// below (one or the other is picked by the size of the level) must agree bit for bit
__device__ inline double cheb_dn(double c1, double d, double c2, double dinv, double b, double y) {
#pragma clang fp contract(off)
    const double t = c2 * (dinv * (b - y));
    return fma(c1, d, t);
}

template <int DOF, int EPI>
__global__ __launch_bounds__(BLK) void k_dia_row(DiaOp<DOF> op, NodeArgs a) {
    const Geom &g = op.g;
    const long plane = g.plane();
    const long tl = blockIdx.x * (long)BLK + threadIdx.x;
    double pdot = 0.0;
    if (tl < op.rows_here()) {
        const long t = op.row_of(tl);
        const long q = t + plane * g.own_lo * DOF;  // row
        const long n = q / DOF;
        const int k = (int)(n / plane);
        const int rem = (int)(n % plane);
        const int j = rem / g.nx, i = rem % g.nx;
        const double *__restrict__ u = a.x;
        double y = 0.0;
        // branch-free: coefficients of non-existent neighbours are stored as zeros, so only the
        // ADDRESS has to be made safe (clamped to the own node); all 2 x 27 x DOF loads can be in flight
#pragma unroll
        for (int dk = -1; dk <= 1; dk++) {
            const bool okk = k + dk >= 0 && k + dk < g.nzl;
#pragma unroll
            for (int dj = -1; dj <= 1; dj++) {
                const bool okj = okk && j + dj >= 0 && j + dj < g.ny;
#pragma unroll
                for (int di = -1; di <= 1; di++) {
                    const bool ok = okj && i + di >= 0 && i + di < g.nx;
                    const int blk = (dk + 1) * 9 + (dj + 1) * 3 + (di + 1);
                    const long nb = ok ? n + di + (long)g.nx * (dj + (long)g.ny * dk) : n;
#pragma unroll
                    for (int c = 0; c < DOF; c++) y = fma(op.S[(long)(blk * DOF + c) * op.nrows + q], u[nb * DOF + c], y);
                }
            }
        }
        if (EPI == EPI_APPLY) {
            a.out[q] = a.dinv ? a.dinv[q] * y : y;
        } else if (EPI == EPI_RESID) {
            a.out[q] = a.b[q] - y;
        } else if (EPI == EPI_CHEB) {
            const double dn = cheb_dn(a.c1, a.d[q], a.c2, a.dinv[q], a.b[q], y);
            a.d[q] = dn;
            a.out[q] = u[q] + dn;
        } else {
            a.out[q] = y;
            pdot = u[q] * y;
        }
    }
    if (EPI == EPI_APPLY_DOT) {
        const double v[1] = {block_sum(pdot)};
        reduce_tail<1>(v, a.partials, gridDim.x, blockIdx.x, a.ticket, a.red_out);
    }
}

// Round 6: what the mirrored reads change, and the correction that goes with them.  Reading the block towards an "upper"
// neighbour from that neighbour's row (transposed) replaces A_ij by A_ji^T.  The stored stencil is symmetric only to the
// rounding of its Galerkin sums, and entry by entry that is harmless -- but it changes every ROW SUM by sum_j (A_ji^T - A_ij): the
// row's answer to a rigid translation, itself only the rounding residue of an exact zero and, as on the fine level,
// what a multigrid iterate with its large smooth component feels.  Measured on C3 (256 x 128 x 128, 14 iterations): the last
// ||r_k|| moves by 6.5e-11 against the CPU checker when level 2 takes the mirrored reads, 1.6e-13 for a mere change of summation
// order.  So the difference is kept: dS = sum over the mirrored neighbours of (A_ij - A_ji^T), a 3 x 3 block per node in the
// three extra slices 81 .. 83 of the stencil array, added to the row's result as dS u_i -- each row answers a translation as
// the stored row does, and only fields that vary from node to node see the mirrored values.
__global__ __launch_bounds__(BLK) void k_dia_sym_fix(Geom g, double *__restrict__ S, long nrows) {
    const long plane = g.plane();
    const long t = blockIdx.x * (long)BLK + threadIdx.x;
    if (t >= g.owned_nodes()) return;
    const int k = g.own_lo + (int)(t / plane);
    const int rem = (int)(t % plane);
    const int j = rem / g.nx, i = rem % g.nx;
    const long n = t + plane * g.own_lo;
    double acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int blk = 14; blk < 27; blk++) {
        const int di = blk % 3 - 1, dj = (blk / 3) % 3 - 1, dk = blk / 9 - 1;
        const bool ok = k + dk >= 0 && k + dk < g.nzl && j + dj >= 0 && j + dj < g.ny && i + di >= 0 && i + di < g.nx;
        if (!(ok && k + dk >= g.own_lo && k + dk <= g.own_hi)) continue;   // (the kernel's own condition for a mirrored read)
        const long nb = n + di + (long)g.nx * (dj + (long)g.ny * dk);
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++)
                acc[r][c] += S[(long)(blk * 3 + c) * nrows + n * 3 + r] - S[(long)((26 - blk) * 3 + r) * nrows + nb * 3 + c];
    }
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) S[(long)(81 + c) * nrows + n * 3 + r] = acc[r][c];
}

// Round 6: the 3-way split with a thread per NODE and z-offset instead of per row and z-offset (DOF = 3): the three rows of a
// node share their neighbour indices, boundary flags and the 27 input values, and the 3 x 3 block towards a neighbour is three
// contiguous triples either way it is stored -- own row: S[(blk 3 + c) nrows + 3 n + (0..2)] over the rows, mirrored:
// S[((26 - blk) 3 + rr) nrows + 3 nb + (0..2)] over the columns.  Counters of the row form at 128^3's level 2 (35 937 nodes):
// 25 FP64 instructions in 446 vector + 210 scalar instructions per wave on 3 888 waves -- index arithmetic; this form runs a third
// of the waves.  One wave = one z-offset (the offset and everything derived from it are scalar), 64 nodes per workgroup of 192
// threads.  Every row is summed in the order of k_dia_row_split<3, EPI, 3, SYM> (neighbours (dj, di), then c; the three partial
// sums in LDS in the order of the z-offsets): the same bits.
// Used for every level beyond the 9-way class (mg.h): on the large ones (C3's, the 256^3 class's and C5's level 2) the mirrored
// reads halve the coefficient stream the unsplit row form used to pull from HBM.
template <int EPI, bool SYM>
__global__ __launch_bounds__(192) void k_dia_node3(DiaOp<3> op, NodeArgs a) {
    constexpr int SPLIT = 3;
    constexpr int NPB = 64;
    typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
    __shared__ double s_part[SPLIT == 3 ? 3 : 1][3][SPLIT == 3 ? NPB : 1];
    const Geom &g = op.g;
    const long plane = g.plane();
    const int part = SPLIT == 3 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0, r = SPLIT == 3 ? (threadIdx.x & 63) : threadIdx.x;
    const int nbk = gridDim.x, x8 = blockIdx.x & 7;
    const int bid = SYM ? x8 * (nbk >> 3) + min(x8, nbk & 7) + (blockIdx.x >> 3) : blockIdx.x;
    const long tl = ((long)bid * NPB + r) * 3;          // first row of the node, counted inside the launch
    const bool valid = tl < op.rows_here();
    const long q0 = (valid ? op.row_of(tl) : 0) + plane * g.own_lo * 3;
    const long n = q0 / 3;
    const double *__restrict__ u = a.x;
    const double *__restrict__ S = op.S;
    double e_b[3] = {0, 0, 0}, e_d[3] = {0, 0, 0}, e_di[3] = {0, 0, 0}, e_u[3] = {0, 0, 0};
    if (valid && part == 0) {
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
            if (EPI == EPI_RESID || EPI == EPI_CHEB) e_b[rr] = a.b[q0 + rr];
            if (EPI == EPI_CHEB) e_d[rr] = a.d[q0 + rr], e_di[rr] = a.dinv[q0 + rr];
            if (EPI == EPI_APPLY && a.dinv) e_di[rr] = a.dinv[q0 + rr];
            if (EPI == EPI_CHEB || EPI == EPI_APPLY_DOT || SYM) e_u[rr] = u[q0 + rr];
        }
    }
    // SYM: the row-sum correction of the mirrored reads (k_dia_sym_fix), requested with the epilogue operands
    double e_ds[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    if (SYM && valid && part == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int rr = 0; rr < 3; rr++) e_ds[rr][c] = S[(long)(81 + c) * op.nrows + q0 + rr];
    }
    double y[3] = {0.0, 0.0, 0.0};
    if (valid) {
        const int k = (int)(n / plane);
        const int rem = (int)(n % plane);
        const int j = rem / g.nx, i = rem % g.nx;
#pragma unroll
        for (int dkk = 0; dkk < (SPLIT == 3 ? 1 : 3); dkk++) {
            const int dk = (SPLIT == 3 ? part : dkk) - 1;
            const bool okk = k + dk >= 0 && k + dk < g.nzl;
            const bool own_k = k + dk >= g.own_lo && k + dk <= g.own_hi;
#pragma unroll
            for (int dj = -1; dj <= 1; dj++) {
                const bool okj = okk && j + dj >= 0 && j + dj < g.ny;
#pragma unroll
                for (int di = -1; di <= 1; di++) {
                    const bool ok = okj && i + di >= 0 && i + di < g.nx;
                    const int blk = (dk + 1) * 9 + (dj + 1) * 3 + (di + 1);      // scalar (SPLIT 3) / compile time (SPLIT 1)
                    const long nb = ok ? n + di + (long)g.nx * (dj + (long)g.ny * dk) : n;
                    const bool mir = SYM && blk > 13 && ok && own_k;
                    const double *__restrict__ un = u + nb * 3;
                    const d2u u01 = *(const d2u *)un;
                    const double uv[3] = {u01.x, u01.y, un[2]};
                    // three contiguous triples: T[kq][0..2]
                    double T[3][3];
#pragma unroll
                    for (int kq = 0; kq < 3; kq++) {
                        const long ad = mir ? (long)((26 - blk) * 3 + kq) * op.nrows + nb * 3 : (long)(blk * 3 + kq) * op.nrows + q0;
                        const d2u t01 = *(const d2u *)(S + ad);
                        T[kq][0] = t01.x;
                        T[kq][1] = t01.y;
                        T[kq][2] = S[ad + 2];
                    }
                    // coefficient of row rr towards column c: own row T[c][rr], mirrored T[rr][c]
#pragma unroll
                    for (int c = 0; c < 3; c++)
#pragma unroll
                        for (int rr = 0; rr < 3; rr++) {
                            const double cf = (rr == c) ? T[c][c] : (mir ? T[rr][c] : T[c][rr]);
                            y[rr] = fma(cf, uv[c], y[rr]);
                        }
                }
            }
        }
    }
    if (SPLIT == 3) {
#pragma unroll
        for (int rr = 0; rr < 3; rr++) s_part[part][rr][r] = y[rr];
        __syncthreads();
    }
    double pdot = 0.0;
    if (valid && part == 0) {
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
            double yy = y[rr];
            if (SPLIT == 3) {
                yy = s_part[0][rr][r];
                yy += s_part[1][rr][r];
                yy += s_part[2][rr][r];
            }
            if (SYM) {
#pragma unroll
                for (int c = 0; c < 3; c++) yy = fma(e_ds[rr][c], e_u[c], yy);
            }
            const long q = q0 + rr;
            if (EPI == EPI_APPLY) {
                a.out[q] = a.dinv ? e_di[rr] * yy : yy;
            } else if (EPI == EPI_RESID) {
                a.out[q] = e_b[rr] - yy;
            } else if (EPI == EPI_CHEB) {
                const double dn = cheb_dn(a.c1, e_d[rr], a.c2, e_di[rr], e_b[rr], yy);
                a.d[q] = dn;
                a.out[q] = e_u[rr] + dn;
            } else {
                a.out[q] = yy;
                pdot = fma(e_u[rr], yy, pdot);
            }
        }
    }
    if (EPI == EPI_APPLY_DOT) {
        const double v[1] = {block_sum(pdot)};
        reduce_tail<1>(v, a.partials, gridDim.x, blockIdx.x, a.ticket, a.red_out);
    }
}

// The same operator for the SMALLEST levels (17^3 nodes: 58 workgroups of rows would leave 3/4 of the chip idle and
// queue 162 loads per thread behind one CU's address unit): a row is split over SPLIT threads -- by z-offset (3) or
// (z, y)-offset (9) of the neighbour -- whose partial sums meet in LDS in a fixed order.
// SYM: the matrix is symmetric, so the coefficient towards a neighbour in the "upper" half of the stencil is also
// stored in that neighbour's own row (opposite offset, transposed block).  Reading it from there makes every stored
// value serve two rows; with the workgroups of an XCD covering a contiguous run of rows both uses meet in one L2.
// Only rows whose neighbour row is owned take the mirrored address (ghost rows are not assembled).
template <int DOF, int EPI, int SPLIT, bool SYM = false>
__global__ __launch_bounds__(BLK) void k_dia_row_split(DiaOp<DOF> op, NodeArgs a) {
    constexpr int RPB = BLK / SPLIT;  // rows per workgroup
    __shared__ double s_part[SPLIT][RPB];
    const Geom &g = op.g;
    const long plane = g.plane();
    const int part = threadIdx.x / RPB, r = threadIdx.x % RPB;
    // workgroups are dealt round-robin to the 8 XCDs: give every XCD a contiguous run of rows
    const int nbk = gridDim.x, x8 = blockIdx.x & 7;
    const int bid = SYM ? x8 * (nbk >> 3) + min(x8, nbk & 7) + (blockIdx.x >> 3) : blockIdx.x;
    const long tl = bid * (long)RPB + r;
    const bool valid = part < SPLIT && tl < op.rows_here();
    const long t = valid ? op.row_of(tl) : 0;
    const long q = t + plane * g.own_lo * DOF;  // row
    const double *__restrict__ u = a.x;
    // epilogue operands of the row: requested BEFORE the stencil loads, not after the barrier (these kernels run at
    // the latency floor on the coarse levels: one memory round trip instead of two)
    double e_b = 0.0, e_d = 0.0, e_di = 0.0, e_u = 0.0;
    if (valid && part == 0) {
        if (EPI == EPI_RESID || EPI == EPI_CHEB) e_b = a.b[q];
        if (EPI == EPI_CHEB) e_d = a.d[q], e_di = a.dinv[q];
        if (EPI == EPI_APPLY && a.dinv) e_di = a.dinv[q];
        if (EPI == EPI_CHEB || EPI == EPI_APPLY_DOT) e_u = u[q];
    }
    // SYM: the row-sum correction of the mirrored reads (k_dia_sym_fix; the same three fma as k_dia_node3, the same bits)
    double e_ds[DOF], e_un[DOF];
#pragma unroll
    for (int c = 0; c < DOF; c++) e_ds[c] = 0.0, e_un[c] = 0.0;
    if (SYM && DOF == 3 && valid && part == 0) {
        const long nn = q / DOF;
#pragma unroll
        for (int c = 0; c < DOF; c++) {
            e_ds[c] = op.S[(long)(27 * DOF + c) * op.nrows + q];
            e_un[c] = u[nn * DOF + c];
        }
    }
    if (valid) {
        const long n = q / DOF;
        const int k = (int)(n / plane);
        const int rem = (int)(n % plane);
        const int j = rem / g.nx, i = rem % g.nx;
        double y = 0.0;
        const int dk = (SPLIT == 3 ? part : part / 3) - 1;
        const bool okk = k + dk >= 0 && k + dk < g.nzl;
#pragma unroll
        for (int djj = 0; djj < (SPLIT == 9 ? 1 : 3); djj++) {
            const int dj = (SPLIT == 9 ? part % 3 : djj) - 1;
            const bool okj = okk && j + dj >= 0 && j + dj < g.ny;
#pragma unroll
            for (int di = -1; di <= 1; di++) {
                const bool ok = okj && i + di >= 0 && i + di < g.nx;
                const int blk = (dk + 1) * 9 + (dj + 1) * 3 + (di + 1);
                const long nb = ok ? n + di + (long)g.nx * (dj + (long)g.ny * dk) : n;
                const bool mir = SYM && blk > 13 && ok && k + dk >= g.own_lo && k + dk <= g.own_hi;
                const int rr = (int)(q - n * DOF);
#pragma unroll
                for (int c = 0; c < DOF; c++) {
                    const long ad = mir ? (long)((26 - blk) * DOF + rr) * op.nrows + nb * DOF + c
                                        : (long)(blk * DOF + c) * op.nrows + q;
                    y = fma(op.S[ad], u[nb * DOF + c], y);
                }
            }
        }
        s_part[part][r] = y;
    }
    __syncthreads();
    double pdot = 0.0;
    if (valid && part == 0) {
        double y = s_part[0][r];
#pragma unroll
        for (int p = 1; p < SPLIT; p++) y += s_part[p][r];
        if (SYM && DOF == 3) {
#pragma unroll
            for (int c = 0; c < DOF; c++) y = fma(e_ds[c], e_un[c], y);
        }
        if (EPI == EPI_APPLY) {
            a.out[q] = a.dinv ? e_di * y : y;
        } else if (EPI == EPI_RESID) {
            a.out[q] = e_b - y;
        } else if (EPI == EPI_CHEB) {
            const double dn = cheb_dn(a.c1, e_d, a.c2, e_di, e_b, y);
            a.d[q] = dn;
            a.out[q] = e_u + dn;
        } else {
            a.out[q] = y;
            pdot = e_u * y;
        }
    }
    if (EPI == EPI_APPLY_DOT) {
        const double v[1] = {block_sum(pdot)};
        reduce_tail<1>(v, a.partials, gridDim.x, blockIdx.x, a.ticket, a.red_out);
    }
}

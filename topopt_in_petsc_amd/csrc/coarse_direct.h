// coarse_direct.h -- the coarsest level solved EXACTLY: explicit triangular inverse of its banded Cholesky factor.
//
// The reference solves the coarsest level with a Krylov method to a tight tolerance (LinearElasticity.cc:720-731); the
// Chebyshev run that stands for it here costs 20 dependent steps per visit (2.3 us each even inside one launch,
// coarse_run.h) and four visits per W-cycle, and its accuracy caps the convergence of the outer CG (13 iterations with
// 20 steps, 11 with 40 or with an exact solve -- tools/cycle_experiment_exact_coarse.py).  A level of a few thousand rows
// (9^3 nodes at 128^3: n = 2187, half bandwidth 275) has another fast form:
//     A = L L^T  (banded, block size 32),   W = L^-1  (dense lower triangle),   A^-1 b = W^T (W b)
// i.e. two triangular matrix-vector products per visit -- no dependent chain, 2 x 19 MB read by the whole device --
// at the price of a factorisation per design iteration, which runs on a stream of its own beside the spectra estimates of
// the other levels:
//   k_cd_fill     stencil (DIA) rows -> block-band storage of the lower triangle;
//   k_cd_factor   left-looking block Cholesky; the KB+1 tiles of a block column are computed by KB+1 workgroups that sit on
//                 ONE XCD (join protocol and barrier of coarse_run.h; two barriers per block column); the workgroup of the
//                 diagonal tile factors it and inverts the 32 x 32 factor, the others multiply with that inverse;
//   k_cd_invert   W = L^-1 by block forward substitution, one workgroup per block of 32 columns, no synchronisation
//                 between workgroups (columns are independent); the last KB row blocks of the workgroup's columns live
//                 in an LDS ring; also writes W^T;
//   k_cd_lower / k_cd_upper   y = W b,  x = W^T y: one wave per row.
// Sums are formed in a fixed order: results are reproducible run to run; against the CPU restatement (oracle: banded
// Cholesky, two substitutions) they differ by rounding only.  A non-positive pivot poisons the factor with NaN, which the
// Krylov loop reports as divergence.
#pragma once
#include "coarse_run.h"

constexpr int CD_NB = 32, CD_T = 256, CD_KBMAX = 16, CD_LD = CD_NB + 1;
constexpr int CD_MAXROWS = 4096;

struct CdGeom {
    int n, np, nblk, KB;  // rows, rows padded to whole blocks, blocks, band width in blocks (below the diagonal)
};
// block (i, j), i - KB <= j <= i, of the band storage: 32 x 32 doubles, row-major
__host__ __device__ inline long cd_blk(const CdGeom &c, int i, int j) { return ((long)i * (c.KB + 1) + (j - (i - c.KB))) * (CD_NB * CD_NB); }

template <int DOF>
__global__ __launch_bounds__(CD_T) void k_cd_fill(DiaOp<DOF> op, CdGeom c, double *__restrict__ Lb) {
    const long q = blockIdx.x * (long)CD_T + threadIdx.x;
    if (q >= c.np) return;
    const int bi = (int)(q / CD_NB), r = (int)(q % CD_NB);
    if (q >= c.n) {  // padding rows: identity
        Lb[cd_blk(c, bi, bi) + r * CD_NB + r] = 1.0;
        return;
    }
    const Geom &g = op.g;
    const long plane = g.plane();
    const long n = q / DOF;
    const int k = (int)(n / plane), rem = (int)(n % plane), j = rem / g.nx, i = rem % g.nx;
    for (int dk = -1; dk <= 1; dk++)
        for (int dj = -1; dj <= 1; dj++)
            for (int di = -1; di <= 1; di++) {
                if (k + dk < 0 || k + dk >= g.nzl || j + dj < 0 || j + dj >= g.ny || i + di < 0 || i + di >= g.nx) continue;
                const long nb = n + di + (long)g.nx * (dj + (long)g.ny * dk);
                const int blk = (dk + 1) * 9 + (dj + 1) * 3 + (di + 1);
                for (int cc = 0; cc < DOF; cc++) {
                    const long col = nb * DOF + cc;
                    if (col > q) continue;
                    Lb[cd_blk(c, bi, (int)(col / CD_NB)) + r * CD_NB + (int)(col % CD_NB)] = op.S[(long)(blk * DOF + cc) * op.nrows + q];
                }
            }
}

// 32 bytes per thread of a 32 x 32 block, past the L1 (the block was written by another workgroup of this kernel); the
// block address is uniform (a descriptor in SGPRs), the thread's place in it the offset
__device__ inline void cd_load4(const double *block, int t, double v[4]) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(block), 0, CD_NB * CD_NB * 8, 0x00020000);
    const u4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, t * 32, 0, 16 /* sc1 */);
    const u4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, t * 32 + 16, 0, 16);
    const d2 x = __builtin_bit_cast(d2, a), y = __builtin_bit_cast(d2, b);
    v[0] = x[0], v[1] = x[1], v[2] = y[0], v[3] = y[1];
}

// reciprocal square root to (almost) full precision without the division / square-root sequences of the compiler
__device__ inline double cd_rsqrt(double d) {
    double y = __builtin_amdgcn_rsq(d);
    y = y * fma(-0.5 * d * y, y, 1.5);
    y = y * fma(-0.5 * d * y, y, 1.5);
    return y;
}

__global__ __launch_bounds__(CD_T) void k_cd_factor(CdGeom c, double *Lb, double *__restrict__ Linv, XcdRunCtrl *ctl, int P) {
    __shared__ double sA[CD_NB][CD_LD], sB[CD_NB][CD_LD], sT[CD_NB][CD_LD], sX[CD_NB][CD_LD];
    __shared__ double s_invd[CD_NB], s_d;
    __shared__ int s_rank, s_dead;
    if (threadIdx.x == 0) s_rank = xcd_join(ctl, P);
    __syncthreads();
    const int rank = s_rank;
    if (rank < 0) {
        xcd_leave(ctl);
        return;
    }
    const int t = threadIdx.x, r = t >> 3, sub = t & 7, c0 = sub * 4;
    int nbar = 0;
    bool dead = false;
    for (int k = 0; k < c.nblk && !dead; k++) {
        const int i = k + rank;
        const bool act = i < c.nblk;
        double acc[4] = {0, 0, 0, 0};
        if (act) {
            const double *At = Lb + cd_blk(c, i, k) + t * 4;  // still the matrix: nobody has written this tile yet
#pragma unroll
            for (int e = 0; e < 4; e++) acc[e] = At[e];
            const int jlo = max(i - c.KB, 0), nj = k - jlo;
            double vA[CD_KBMAX][4], vB[CD_KBMAX][4];
#pragma unroll
            for (int jj = 0; jj < CD_KBMAX; jj++)
                if (jj < nj) {
                    cd_load4(Lb + cd_blk(c, i, jlo + jj), t, vA[jj]);
                    cd_load4(Lb + cd_blk(c, k, jlo + jj), t, vB[jj]);
                }
#pragma unroll
            for (int jj = 0; jj < CD_KBMAX; jj++)
                if (jj < nj) {
                    __syncthreads();
#pragma unroll
                    for (int e = 0; e < 4; e++) sA[r][c0 + e] = vA[jj][e], sB[r][c0 + e] = vB[jj][e];
                    __syncthreads();
#pragma unroll 8
                    for (int m = 0; m < CD_NB; m++) {
                        const double a = sA[r][m];
#pragma unroll
                        for (int e = 0; e < 4; e++) acc[e] = fma(-a, sB[c0 + e][m], acc[e]);
                    }
                }
        }
        if (rank == 0) {  // the diagonal tile: L_kk L_kk^T = T (lower), then its inverse
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; e++) sT[r][c0 + e] = acc[e];
            __syncthreads();
            for (int cc = 0; cc < CD_NB; cc++) {
                double s = 0.0;
                for (int m = sub; m < cc; m += 8) s = fma(sT[r][m], sT[cc][m], s);
                s += __shfl_xor(s, 1);
                s += __shfl_xor(s, 2);
                s += __shfl_xor(s, 4);
                const double a = sT[r][cc] - s;
                if (r == cc && sub == 0) s_d = a;
                __syncthreads();
                const double d = s_d;
                const double y = d > 0.0 ? cd_rsqrt(d) : __builtin_nan("");
                if (sub == 0) {
                    if (r == cc) sT[r][cc] = d * y, s_invd[cc] = y;
                    else if (r > cc) sT[r][cc] = a * y;
                    else sT[r][cc] = 0.0;
                }
                __syncthreads();
            }
            // X = L_kk^-1, column e = r of this thread group (8 threads per column), rows in sequence
            const int ecol = r;
            for (int rr = 0; rr < CD_NB; rr++) {
                double s = 0.0;
                for (int m = ecol + sub; m < rr; m += 8) s = fma(sT[rr][m], sX[m][ecol], s);
                s += __shfl_xor(s, 1);
                s += __shfl_xor(s, 2);
                s += __shfl_xor(s, 4);
                if (sub == 0) sX[rr][ecol] = rr < ecol ? 0.0 : ((rr == ecol ? 1.0 : 0.0) - s) * s_invd[rr];
                __syncthreads();
            }
            double *Lt = Lb + cd_blk(c, k, k) + t * 4, *Xt = Linv + (long)k * (CD_NB * CD_NB) + t * 4;
#pragma unroll
            for (int e = 0; e < 4; e++) Lt[e] = sT[r][c0 + e], Xt[e] = sX[r][c0 + e];
        }
        if (xcd_barrier(ctl, nbar++, P, &s_dead)) {
            dead = true;
            break;
        }
        if (act && rank > 0) {  // L_ik = T L_kk^-T
            double vX[4];
            cd_load4(Linv + (long)k * (CD_NB * CD_NB), t, vX);
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; e++) sT[r][c0 + e] = acc[e], sB[r][c0 + e] = vX[e];
            __syncthreads();
            double o[4] = {0, 0, 0, 0};
#pragma unroll 8
            for (int m = 0; m < CD_NB; m++) {
                const double a = sT[r][m];
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = fma(a, sB[c0 + e][m], o[e]);
            }
            double *Lt = Lb + cd_blk(c, i, k) + t * 4;
#pragma unroll
            for (int e = 0; e < 4; e++) Lt[e] = o[e];
        }
        if (xcd_barrier(ctl, nbar++, P, &s_dead)) {
            dead = true;
            break;
        }
    }
    if (dead && rank == 0 && threadIdx.x == 0) Linv[0] = __builtin_nan("");
    xcd_leave(ctl);
}

// W = L^-1 (and its transpose), block column `cb` per workgroup
__global__ __launch_bounds__(CD_T) void k_cd_invert(CdGeom c, const double *__restrict__ Lb, const double *__restrict__ Linv,
                                                    double *__restrict__ W, double *__restrict__ Wt) {
    __shared__ double sW[CD_KBMAX][CD_NB][CD_NB];  // ring: row blocks i - KB .. i - 1 of this block column
    __shared__ double sA[CD_NB][CD_LD], sT[CD_NB][CD_LD], sI[CD_NB][CD_LD];
    const int cb = blockIdx.x, t = threadIdx.x, r = t >> 3, sub = t & 7, c0 = sub * 4;
    const int ring = c.KB;
    for (int i = cb; i < c.nblk; i++) {
        const int jlo = max(i - c.KB, cb), nj = i - jlo;
        double vA[CD_KBMAX][4], vI[4];
#pragma unroll
        for (int jj = 0; jj < CD_KBMAX; jj++)
            if (jj < nj) {
                const double *p = Lb + cd_blk(c, i, jlo + jj) + t * 4;
#pragma unroll
                for (int e = 0; e < 4; e++) vA[jj][e] = p[e];
            }
        {
            const double *p = Linv + (long)i * (CD_NB * CD_NB) + t * 4;
#pragma unroll
            for (int e = 0; e < 4; e++) vI[e] = p[e];
        }
        double acc[4];
#pragma unroll
        for (int e = 0; e < 4; e++) acc[e] = (i == cb && r == c0 + e) ? 1.0 : 0.0;
#pragma unroll
        for (int jj = 0; jj < CD_KBMAX; jj++)
            if (jj < nj) {
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 4; e++) sA[r][c0 + e] = vA[jj][e];
                __syncthreads();
                const double(*Wj)[CD_NB] = sW[(jlo + jj) % ring];
#pragma unroll 8
                for (int m = 0; m < CD_NB; m++) {
                    const double a = sA[r][m];
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[e] = fma(-a, Wj[m][c0 + e], acc[e]);
                }
            }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; e++) sT[r][c0 + e] = acc[e], sI[r][c0 + e] = vI[e];
        __syncthreads();
        double o[4] = {0, 0, 0, 0};
#pragma unroll 8
        for (int m = 0; m < CD_NB; m++) {
            const double a = sI[r][m];
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = fma(a, sT[m][c0 + e], o[e]);
        }
        __syncthreads();  // every read of the slot that is overwritten now (block i - KB) and of sT is done
        double(*Wi)[CD_NB] = sW[i % ring];
        double *wp = W + (long)(i * CD_NB + r) * c.np + cb * CD_NB + c0;
#pragma unroll
        for (int e = 0; e < 4; e++) Wi[r][c0 + e] = o[e], sT[r][c0 + e] = o[e], wp[e] = o[e];
        __syncthreads();
        double *tp = Wt + (long)(cb * CD_NB + r) * c.np + i * CD_NB + c0;  // row r of the transposed tile
#pragma unroll
        for (int e = 0; e < 4; e++) tp[e] = sT[c0 + e][r];
    }
}

// y[i] = sum_{j <= i} W[i][j] b[j]   (UPPER: x[i] = sum_{j >= i} Wt[i][j] y[j]); one wave per row, fixed order
template <bool UPPER>
__global__ __launch_bounds__(CD_T) void k_cd_tri(CdGeom c, const double *__restrict__ M, const double *__restrict__ v, double *__restrict__ out) {
    const int row = blockIdx.x * (CD_T / WAVE) + threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
    if (row >= c.n) return;
    const double *__restrict__ mr = M + (long)row * c.np;
    const int lo = UPPER ? (row & ~(CD_NB - 1)) : 0, hi = UPPER ? c.n : row + 1;  // (the diagonal tile holds zeros below the diagonal)
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int j = lo + lane;
    for (; j + 3 * WAVE < hi; j += 4 * WAVE) {
        s0 = fma(mr[j], v[j], s0);
        s1 = fma(mr[j + WAVE], v[j + WAVE], s1);
        s2 = fma(mr[j + 2 * WAVE], v[j + 2 * WAVE], s2);
        s3 = fma(mr[j + 3 * WAVE], v[j + 3 * WAVE], s3);
    }
    for (; j < hi; j += WAVE) s0 = fma(mr[j], v[j], s0);
    const double s = wave_sum((s0 + s1) + (s2 + s3));
    if (lane == 0) out[row] = s;
}

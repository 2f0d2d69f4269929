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
//   k_cd_fill     stencil (DIA) rows -> block-band storage of the lower triangle (32 x 32 blocks, stored transposed);
//   k_cd_factor   block Cholesky, left-looking per tile (round 4: the tile substitution of stage (C) on all four waves).  KB+1 workgroups on ONE XCD (join protocol and barrier of
//                 coarse_run.h); a workgroup OWNS a row block (i -> workgroup i mod (KB+1)) and walks its tiles left to right
//                 with the row block's finished tiles in LDS.  Per block column k: the owner of row block k finishes the
//                 diagonal tile (one product of its own data) and factors it in one wave (registers + LDS, no workgroup
//                 barrier inside); meanwhile the others pre-compute their NEXT tile's sum over the columns already
//                 published (look-ahead: this is where the flops are, off the critical path); barrier; the others solve
//                 their tile against the 32 x 32 factor (one wave, row per lane) and publish it; barrier.
//   k_cd_diag_inv the 32 x 32 diagonal factors inverted (one wave per block);
//   k_cd_dc_*     (round 4, default) W = L^-1 and W^T by divide and conquer over block ranges: 7 levels of batched tile products;
//   k_cd_invert   (TP_CD_INVERT_COLUMNS=1) W = L^-1 by block forward substitution, one workgroup per block of 32 columns, no synchronisation
//                 between workgroups (columns are independent); the last KB row blocks of the workgroup's columns live
//                 in an LDS ring; also writes W^T;
//   k_cd_tri      y = W b,  x = W^T y: one wave per row.
// The 32 x 32 x 32 products run one per WAVE on 4 x 4 register tiles (operands transposed in LDS, 16-byte reads: LDS
// bandwidth is what bounds them), the terms of a sum are dealt to the four waves and added up in a fixed order.
// Sums are formed in a fixed order: results are reproducible run to run; against a CPU restatement (tests: banded
// Cholesky, two substitutions) they differ by rounding only.  A non-positive pivot poisons the factor with NaN, which the
// Krylov loop reports as divergence.
#pragma once
#include "coarse_run.h"

constexpr int CD_NB = 32, CD_T = 256, CD_KBMAX = 12, CD_LD = CD_NB + 1, CD_BLK = CD_NB * CD_NB;
constexpr int CD_MAXROWS = 4096;

struct CdGeom {
    int n, np, nblk, KB;  // rows, rows padded to whole blocks, blocks, band width in blocks (below the diagonal)
};
// block (i, j), i - KB <= j <= i, of the band storage: 32 x 32 doubles, TRANSPOSED: element (r, m) at [m * 32 + r]
__host__ __device__ inline long cd_blk(const CdGeom &c, int i, int j) { return ((long)i * (c.KB + 1) + (j - (i - c.KB))) * CD_BLK; }

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
                    Lb[cd_blk(c, bi, (int)(col / CD_NB)) + (int)(col % CD_NB) * CD_NB + r] = op.S[(long)(blk * DOF + cc) * op.nrows + q];
                }
            }
}

typedef double cd_d2 __attribute__((ext_vector_type(2)));
typedef unsigned cd_u4 __attribute__((ext_vector_type(4)));

// one wave copies a published 32 x 32 block (8 KB) into LDS, past the L1: the block was written by another workgroup of
// this kernel.  The block address is uniform (descriptor in SGPRs), the lane's place in it the offset.
__device__ inline void cd_stage_block(const double *block, double *lds, int lane) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(block), 0, CD_BLK * 8, 0x00020000);
    cd_u4 v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (q * WAVE + lane) * 16, 0, 16 /* sc1 */);
#pragma unroll
    for (int q = 0; q < 8; q++) reinterpret_cast<cd_u4 *>(lds)[q * WAVE + lane] = v[q];
}
// one wave: acc[a][b] += sum_m At[m][4 tr + a] * Bt[m][4 tc + b], lane = 8 tr + tc  (both operands [m][32] in LDS)
__device__ inline void cd_gemm_nt(double (&acc)[16], const double *At, const double *Bt, int lane) {
    const int tr = lane >> 3, tc = lane & 7;
#pragma unroll 4
    for (int m = 0; m < CD_NB; m++) {
        const cd_d2 a01 = *reinterpret_cast<const cd_d2 *>(At + m * CD_NB + 4 * tr), a23 = *reinterpret_cast<const cd_d2 *>(At + m * CD_NB + 4 * tr + 2);
        const cd_d2 b01 = *reinterpret_cast<const cd_d2 *>(Bt + m * CD_NB + 4 * tc), b23 = *reinterpret_cast<const cd_d2 *>(Bt + m * CD_NB + 4 * tc + 2);
        const double a[4] = {a01[0], a01[1], a23[0], a23[1]}, b[4] = {b01[0], b01[1], b23[0], b23[1]};
#pragma unroll
        for (int x = 0; x < 4; x++)
#pragma unroll
            for (int y = 0; y < 4; y++) acc[x * 4 + y] = fma(a[x], b[y], acc[x * 4 + y]);
    }
}

// reciprocal square root to (almost) full precision without the division / square-root sequences of the compiler
__device__ inline double cd_rsqrt(double d) {
    double y = __builtin_amdgcn_rsq(d);
    y = y * fma(-0.5 * d * y, y, 1.5);
    y = y * fma(-0.5 * d * y, y, 1.5);
    return y;
}
__device__ inline double cd_readlane(double v, int l) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, l), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// 32 steps of the substitution x L^T = t on a row of a tile held by 16 lanes (two columns per lane, c0 = 2 * (lane & 15)):
// step M: x_M = acc_M / l_MM in its owner (lane M / 2 of the row), handed to the row's lanes by a DPP row_share, then
// acc_c -= x_M l_cM for c > M.  sDt[m][c] = l_cm with the reciprocal of the diagonal on the diagonal.  (The DPP control is an
// immediate: the steps are a compile-time recursion.)
template <int M>
__device__ __forceinline__ void cd_sub_steps(double &x0, double &x1, const double *sDt, int c0, int q16) {
    if constexpr (M < CD_NB) {
        const cd_d2 lm = *reinterpret_cast<const cd_d2 *>(sDt + M * CD_NB + c0);  // l_{c0 M}, l_{c0+1 M}
        const double dm = sDt[M * CD_NB + M];                                    // 1 / l_MM
        const double xm = ((M & 1) ? x1 : x0) * dm;
        if (q16 == (M >> 1)) {
            if (M & 1) x1 = xm;
            else x0 = xm;
        }
        const unsigned long long u = __builtin_bit_cast(unsigned long long, xm);
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x150 + (M >> 1), 0xf, 0xf, false);
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x150 + (M >> 1), 0xf, 0xf, false);
        const double xb = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);  // x_M of this row, in all its lanes
        // columns at or left of M are finished: weight 0 (fma(-x, 0, acc) = acc)
        x0 = fma(-xb, c0 > M ? lm[0] : 0.0, x0);
        x1 = fma(-xb, c0 + 1 > M ? lm[1] : 0.0, x1);
        cd_sub_steps<M + 1>(x0, x1, sDt, c0, q16);
    }
}

// Ld: per block column the 32 x 32 diagonal factor, row-major, with the RECIPROCALS of its diagonal on the diagonal
__global__ __launch_bounds__(CD_T) void k_cd_factor(CdGeom c, double *Lb, double *__restrict__ Ld, XcdRunCtrl *ctl, int P, long long *prof) {
    __shared__ double sOwn[CD_KBMAX * CD_BLK];  // finished tiles of my row block, transposed, slot j % KB
    __shared__ double sStage[4 * CD_BLK];       // per wave: the other operand of a product / the wave's partial tile
    __shared__ double sT[CD_NB * CD_LD], sTn[CD_NB * CD_LD], sD[CD_NB * CD_LD];  // this tile, the next one (look-ahead), diagonal factor
    __shared__ int s_rank, s_dead;
    if (threadIdx.x == 0) s_rank = xcd_join(ctl, P);
    __syncthreads();
    const int rank = s_rank;
    if (rank < 0) {
        xcd_leave(ctl);
        return;
    }
    const int t = threadIdx.x, wave = t / WAVE, lane = t & (WAVE - 1);
    const int tr = lane >> 3, tc = lane & 7;
    int i = rank;  // my row block; its first tile is (i, max(i - KB, 0)) = (i, 0)
    // thread's 4 elements of a transposed global block: column m_g, rows r_g .. r_g + 3
    const int m_g = t >> 3, r_g = (t & 7) * 4;
    auto load_matrix_tile = [&](int bi, int bj) {  // A tile (bi, bj) -> sTn (row-major, padded)
        const double *p = Lb + cd_blk(c, bi, bj) + t * 4;
#pragma unroll
        for (int e = 0; e < 4; e++) sTn[(r_g + e) * CD_LD + m_g] = p[e];
    };
    if (i < c.nblk) load_matrix_tile(i, 0);
    __syncthreads();
    int nbar = 0;
    bool dead = false;
    long long tp0 = prof ? wall_clock64() : 0;
    auto mark = [&](int ph) {  // (timing aid, TP_CD_PROF=1: wall-clock ticks per phase and workgroup)
        if (prof) {
            const long long now = wall_clock64();
            if (threadIdx.x == 0) prof[rank * 8 + ph] += now - tp0;
            tp0 = now;
        }
    };
    for (int k = 0; k < c.nblk; k++) {
        const bool act = i < c.nblk;
        const bool diag = act && i == k;
        mark(7);
        // ---- (A) finish tile (i, k): the product with block column k - 1 is the one the look-ahead could not have
        if (wave == 0 && act) {
            double acc[16];
#pragma unroll
            for (int x = 0; x < 16; x++) acc[x] = 0.0;
            if (k >= 1 && k - 1 >= i - c.KB) {
                const double *At = sOwn + ((k - 1) % c.KB) * CD_BLK, *Bt = At;
                if (!diag) {
                    cd_stage_block(Lb + cd_blk(c, k, k - 1), sStage, lane);
                    Bt = sStage;
                }
                cd_gemm_nt(acc, At, Bt, lane);
            }
#pragma unroll
            for (int x = 0; x < 4; x++)
#pragma unroll
                for (int y = 0; y < 4; y++) sT[(4 * tr + x) * CD_LD + 4 * tc + y] = sTn[(4 * tr + x) * CD_LD + 4 * tc + y] - acc[x * 4 + y];
        }
        __syncthreads();
        mark(0);
        // ---- (B) diagonal owner: factor the tile (wave 0); the others: look-ahead, the sum of tile (i, k + 1) over the
        // block columns published so far (j <= k - 1), terms dealt to the waves
        const bool la = act && !diag && k + 1 <= i;
        if (la) {
            double acc[16];
#pragma unroll
            for (int x = 0; x < 16; x++) acc[x] = 0.0;
            const int jlo = max(i - c.KB, 0);
            for (int j = jlo + wave; j <= k - 1; j += 4) {
                const double *At = sOwn + (j % c.KB) * CD_BLK, *Bt = At;  // (i == k + 1: the next tile is my diagonal one)
                if (i != k + 1) {
                    cd_stage_block(Lb + cd_blk(c, k + 1, j), sStage + wave * CD_BLK, lane);
                    Bt = sStage + wave * CD_BLK;
                }
                cd_gemm_nt(acc, At, Bt, lane);
            }
#pragma unroll
            for (int x = 0; x < 4; x++)
#pragma unroll
                for (int y = 0; y < 4; y++) sStage[wave * CD_BLK + (4 * tr + x) * CD_NB + 4 * tc + y] = acc[x * 4 + y];
        }
        if (diag && wave == 0) {
            if (lane < CD_NB) {
                // Right-looking, a row per lane in registers.  Column cc: pivot d = a[cc] of lane cc (readlane), y = 1/sqrt(d),
                // l = a[cc] y.  The update of the NEXT column (the only one the next pivot waits for) takes its factor from a
                // readlane; the updates of the columns behind it read the column through LDS (one row of the transposed
                // factor, 16-byte broadcast reads) and are applied one column late, their latency behind the pivot arithmetic.
                const int r = lane;
                double a[CD_NB];
#pragma unroll
                for (int cc = 0; cc < CD_NB; cc++) a[cc] = sT[r * CD_LD + cc];
                double lprev = 0.0;
#pragma unroll
                for (int cc = 0; cc < CD_NB; cc++) {
                    if (cc >= 1) {  // deferred updates with column cc - 1: columns cc + 1 .. 31 (column cc was done by readlane)
#pragma unroll
                        for (int c2 = cc + 1; c2 < CD_NB; c2++) a[c2] = fma(-lprev, sStage[(cc - 1) * CD_NB + c2], a[c2]);
                    }
                    const double d = cd_readlane(a[cc], cc);
                    const double y = d > 0.0 ? cd_rsqrt(d) : __builtin_nan("");
                    const double l = a[cc] * y;
                    sStage[cc * CD_NB + r] = l;  // column cc of the factor, as a row
                    if (cc + 1 < CD_NB) a[cc + 1] = fma(-l, cd_readlane(l, cc + 1), a[cc + 1]);
                    sD[r * CD_LD + cc] = r == cc ? y : (r > cc ? l : 0.0);
                    lprev = l;
                }
            }
            // publish (row-major)
#pragma unroll
            for (int q = 0; q < CD_BLK / WAVE; q++) {
                const int idx = q * WAVE + lane;
                Ld[(long)k * CD_BLK + idx] = sD[(idx >> 5) * CD_LD + (idx & 31)];
            }
        }
        __syncthreads();
        if (la) {  // tile (i, k + 1) so far = matrix tile - partial sums (fixed order)
            const double *p = Lb + cd_blk(c, i, k + 1) + t * 4;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int idx = (r_g + e) * CD_NB + m_g;
                const double sum = (sStage[idx] + sStage[CD_BLK + idx]) + (sStage[2 * CD_BLK + idx] + sStage[3 * CD_BLK + idx]);
                sTn[(r_g + e) * CD_LD + m_g] = p[e] - sum;
            }
        }
        mark(diag ? 1 : 2);
        if (xcd_barrier(ctl, nbar++, P, &s_dead)) {
            dead = true;
            break;
        }
        mark(3);
        // ---- (C) the others: L_ik = T L_kk^-T, into my LDS and out; the owner of k: its next row block.
        // Round 4: the substitution runs on all four waves -- 16 lanes per row of the tile, two columns per lane, the finished
        // x_m handed to the row's lanes by a DPP row_share -- where round 3 ran it as a row per lane on half of ONE wave (496
        // dependent-ish fma per lane behind LDS broadcasts: the longer of the kernel's two single-wave stages, 9.5 of the
        // 28 us of a block column).  Same operations on every entry in the same order: same bits.
        if (act && !diag) {  // (workgroup-uniform)
            double *sDt = sStage;  // (the staging areas are free between the products of (B) and (A))
            {   // the factor of block column k (row-major, reciprocal diagonal) -> transposed: sDt[m][c] = l_cm; all threads
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Ld + (long)k * CD_BLK, 0, CD_BLK * 8, 0x00020000);
                cd_u4 v[2];
#pragma unroll
                for (int q = 0; q < 2; q++) v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (q * CD_T + t) * 16, 0, 16 /* sc1 */);
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int idx = (q * CD_T + t) * 2, rr = idx >> 5, cc = idx & 31;
                    const cd_d2 d = __builtin_bit_cast(cd_d2, v[q]);
                    sDt[cc * CD_NB + rr] = d[0];
                    sDt[(cc + 1) * CD_NB + rr] = d[1];
                }
            }
            __syncthreads();
            double *own = sOwn + (k % c.KB) * CD_BLK, *out = Lb + cd_blk(c, i, k);
            const int q16 = lane & 15, c0 = 2 * q16;
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
                const int r = pass * 16 + wave * 4 + (lane >> 4);
                double x0 = sT[r * CD_LD + c0], x1 = sT[r * CD_LD + c0 + 1];
                cd_sub_steps<0>(x0, x1, sDt, c0, q16);
                own[c0 * CD_NB + r] = x0, own[(c0 + 1) * CD_NB + r] = x1;
                out[c0 * CD_NB + r] = x0, out[(c0 + 1) * CD_NB + r] = x1;
            }
        }
        if (diag) {
            i = k + P;  // first tile of the new row block: (i, k + 1), no products yet
            if (i < c.nblk) load_matrix_tile(i, k + 1);
        }
        mark(4);
        if (xcd_barrier(ctl, nbar++, P, &s_dead)) {
            dead = true;
            break;
        }
        mark(5);
    }
    if (dead && rank == 0 && threadIdx.x == 0) Ld[0] = __builtin_nan("");
    xcd_leave(ctl);
}

// Linv (stored transposed: element (r, m) at [m * 32 + r]) = inverse of the diagonal factor of block k; one wave per block
__global__ __launch_bounds__(WAVE) void k_cd_diag_inv(const double *__restrict__ Ld, double *__restrict__ Linv) {
    __shared__ double sD[CD_NB * CD_LD];
    const int k = blockIdx.x, lane = threadIdx.x;
    for (int idx = lane; idx < CD_BLK; idx += WAVE) sD[(idx >> 5) * CD_LD + (idx & 31)] = Ld[(long)k * CD_BLK + idx];
    __syncthreads();
    if (lane < CD_NB) {
        const int e = lane;  // column e of the inverse
        double x[CD_NB];
#pragma unroll
        for (int r = 0; r < CD_NB; r++) {
            double a0 = r == e ? 1.0 : 0.0, a1 = 0.0;
#pragma unroll
            for (int m = 0; m + 1 < r; m += 2) {
                a0 = fma(-sD[r * CD_LD + m], x[m], a0);
                a1 = fma(-sD[r * CD_LD + m + 1], x[m + 1], a1);
            }
            if (r & 1) a0 = fma(-sD[r * CD_LD + r - 1], x[r - 1], a0);
            x[r] = r < e ? 0.0 : (a0 + a1) * sD[r * CD_LD + r];  // (the diagonal holds reciprocals)
        }
#pragma unroll
        for (int r = 0; r < CD_NB; r++) Linv[(long)k * CD_BLK + e * CD_NB + r] = x[r];
    }
}

// W = L^-1 (and its transpose), block column `cb` per workgroup
__global__ __launch_bounds__(CD_T) void k_cd_invert(CdGeom c, const double *__restrict__ Lb, const double *__restrict__ Linv,
                                                    double *__restrict__ W, double *__restrict__ Wt) {
    __shared__ double sW[CD_KBMAX * CD_BLK];  // ring: row blocks i - KB .. i - 1 of this block column, [m][col]
    __shared__ double sStage[4 * CD_BLK];     // per wave: L_ij transposed / the wave's partial tile
    __shared__ double sT[CD_BLK];             // the right-hand side tile [m][col]
    __shared__ double sO[CD_NB * CD_LD];      // the new row block of W (padded: read by columns for W^T)
    const int cb = blockIdx.x, t = threadIdx.x, wave = t / WAVE, lane = t & (WAVE - 1);
    const int tr = lane >> 3, tc = lane & 7;
    const int ring = c.KB;
    // the factor blocks this wave multiplies with are known in advance (they do not depend on W): the next one is
    // requested before the current product starts, across the row-block steps as well
    cd_u4 pre[8];
    auto first_term = [&](int i) { return max(i - c.KB, cb) + wave; };
    auto request = [&](int i, int j) {
        const cd_u4 *src = reinterpret_cast<const cd_u4 *>(Lb + cd_blk(c, i, j));
#pragma unroll
        for (int q = 0; q < 8; q++) pre[q] = src[q * WAVE + lane];
    };
    {
        int i0 = cb, j0 = first_term(cb);
        while (i0 < c.nblk && j0 >= i0) {  // first (row block, term) of this wave
            i0++;
            if (i0 < c.nblk) j0 = first_term(i0);
        }
        if (i0 < c.nblk) request(i0, j0);
    }
    for (int i = cb; i < c.nblk; i++) {
        cd_u4 li[8];  // wave 0: the inverse of this row block's diagonal factor, requested a step's work ahead of its use
        if (wave == 0) {
            const cd_u4 *src = reinterpret_cast<const cd_u4 *>(Linv + (long)i * CD_BLK);
#pragma unroll
            for (int q = 0; q < 8; q++) li[q] = src[q * WAVE + lane];
        }
        double acc[16];
#pragma unroll
        for (int x = 0; x < 16; x++) acc[x] = 0.0;
        for (int j = first_term(i); j < i; j += 4) {
            double *At = sStage + wave * CD_BLK;
#pragma unroll
            for (int q = 0; q < 8; q++) reinterpret_cast<cd_u4 *>(At)[q * WAVE + lane] = pre[q];
            {   // the wave's next term: in this row block, or the first one of a later row block
                int in = i, jn = j + 4;
                while (in < c.nblk && jn >= in) {
                    in++;
                    if (in < c.nblk) jn = first_term(in);
                }
                if (in < c.nblk) request(in, jn);
            }
            cd_gemm_nt(acc, At, sW + (j % ring) * CD_BLK, lane);
        }
#pragma unroll
        for (int x = 0; x < 4; x++)
#pragma unroll
            for (int y = 0; y < 4; y++) sStage[wave * CD_BLK + (4 * tr + x) * CD_NB + 4 * tc + y] = acc[x * 4 + y];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int idx = t * 4 + e, r = idx >> 5, col = idx & 31;
            const double sum = (sStage[idx] + sStage[CD_BLK + idx]) + (sStage[2 * CD_BLK + idx] + sStage[3 * CD_BLK + idx]);
            sT[idx] = ((i == cb && r == col) ? 1.0 : 0.0) - sum;
        }
        __syncthreads();
        if (wave == 0) {  // W_i = Linv_ii T
            double *At = sStage;
#pragma unroll
            for (int q = 0; q < 8; q++) reinterpret_cast<cd_u4 *>(At)[q * WAVE + lane] = li[q];
            double o[16];
#pragma unroll
            for (int x = 0; x < 16; x++) o[x] = 0.0;
            cd_gemm_nt(o, At, sT, lane);
            double *Wi = sW + (i % ring) * CD_BLK;  // (block i - KB: every product of this step is done)
#pragma unroll
            for (int x = 0; x < 4; x++)
#pragma unroll
                for (int y = 0; y < 4; y++) {
                    Wi[(4 * tr + x) * CD_NB + 4 * tc + y] = o[x * 4 + y];
                    sO[(4 * tr + x) * CD_LD + 4 * tc + y] = o[x * 4 + y];
                }
        }
        __syncthreads();
        {
            const int r = t >> 3, c0 = (t & 7) * 4;
            double *wp = W + (long)(i * CD_NB + r) * c.np + cb * CD_NB + c0;
            double *tp = Wt + (long)(cb * CD_NB + r) * c.np + i * CD_NB + c0;  // row r of the transposed tile
#pragma unroll
            for (int e = 0; e < 4; e++) wp[e] = sO[r * CD_LD + c0 + e], tp[e] = sO[(c0 + e) * CD_LD + r];
        }
        __syncthreads();
    }
}

// ---- W = L^-1 by divide and conquer over block ranges (round 4).  k_cd_invert walks a block column top to bottom: 69
// dependent row steps of ~8 us for the first column, 0.56 ms, most of the chip idle.  For L = [[L11, 0], [L21, L22]]
//     L^-1 = [[W11, 0], [-W22 L21 W11, W22]]:
// the two halves are independent and the coupling is two products, so the work is log2(69) = 7 levels of batched 32 x 32
// tile products, one wave per output tile, every level two launches -- T = L21 W11 (L21 is the band's corner: only the first
// KB block rows of the lower half have entries), then W21 = -W22 T (only the KB block columns of W22 that meet T's rows).
// W and W^T are both kept current (a tile of W^T is the A operand of the second product as it lies in memory).
// Segment of level lv (1 ..): blocks [lo, lo + 2^lv), split at mid = lo + 2^(lv-1); nothing to do where mid >= nblk.
// a 32 x 32 row-major tile (pitch in doubles) on its way to LDS as [m][c]: requested into registers a term ahead of its use
__device__ inline void cd_req_rows(const double *src, long pitch, cd_u4 (&v)[8], int lane) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int idx = q * WAVE + lane, m = idx >> 4, c2 = (idx & 15) * 2;
        v[q] = *reinterpret_cast<const cd_u4 *>(src + (long)m * pitch + c2);
    }
}
__device__ inline void cd_put(double *lds, const cd_u4 (&v)[8], int lane) {  // (idx -> [m][c2]: the linear order of the tile)
#pragma unroll
    for (int q = 0; q < 8; q++) reinterpret_cast<cd_u4 *>(lds)[q * WAVE + lane] = v[q];
}
// level 0: the diagonal tiles, W_ii = Linv_i (stored transposed: element (r, m) at [m * 32 + r])
__global__ __launch_bounds__(CD_T) void k_cd_dc_diag(CdGeom c, const double *__restrict__ Linv, double *__restrict__ W, double *__restrict__ Wt) {
    const int i = blockIdx.x, t = threadIdx.x;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int idx = t * 4 + e, r = idx >> 5, col = idx & 31;
        const double v = Linv[(long)i * CD_BLK + col * CD_NB + r];  // W(r, col)
        W[(long)(i * CD_NB + r) * c.np + i * CD_NB + col] = v;
        Wt[(long)(i * CD_NB + col) * c.np + i * CD_NB + r] = v;
    }
}
// T_(i, jc) = sum_j L_ij W_(j, jc),  i in [mid, mid + KB) (block row y of the corner), jc in [lo, mid) (x), segment z
__global__ __launch_bounds__(WAVE) void k_cd_dc_t(CdGeom c, int lv, const double *__restrict__ Lb, const double *__restrict__ W, double *__restrict__ T) {
    __shared__ double sA[CD_BLK], sB[CD_BLK];
    const int half = 1 << (lv - 1), lo = (int)blockIdx.z << lv, mid = lo + half, hi = min(lo + 2 * half, c.nblk);
    const int i = mid + (int)blockIdx.y, jc = lo + (int)blockIdx.x, lane = threadIdx.x;
    if (mid >= c.nblk || i >= hi || i >= mid + c.KB) return;
    double acc[16];
#pragma unroll
    for (int x = 0; x < 16; x++) acc[x] = 0.0;
    cd_u4 pa[8], pb[8];
    const int j0 = max(max(lo, i - c.KB), jc);  // (W_(j, jc) = 0 above the diagonal)
    auto request = [&](int j) {
        const cd_u4 *la = reinterpret_cast<const cd_u4 *>(Lb + cd_blk(c, i, j));  // [m][a] = L_ij(a, m): the A operand as stored
#pragma unroll
        for (int q = 0; q < 8; q++) pa[q] = la[q * WAVE + lane];
        cd_req_rows(W + (long)j * CD_NB * c.np + jc * CD_NB, c.np, pb, lane);
    };
    request(j0);
    for (int j = j0; j < mid; j++) {
        cd_put(sA, pa, lane);
        cd_put(sB, pb, lane);
        if (j + 1 < mid) request(j + 1);
        __syncthreads();
        cd_gemm_nt(acc, sA, sB, lane);
        __syncthreads();
    }
    const int tr = lane >> 3, tc = lane & 7;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        double *dst = T + (long)(i * CD_NB + 4 * tr + x) * c.np + jc * CD_NB + 4 * tc;
        *reinterpret_cast<cd_d2 *>(dst) = cd_d2{acc[x * 4 + 0], acc[x * 4 + 1]};
        *reinterpret_cast<cd_d2 *>(dst + 2) = cd_d2{acc[x * 4 + 2], acc[x * 4 + 3]};
    }
}
// W_(i, jc) = - sum_k W_(i, k) T_(k, jc),  i in [mid, hi) (y), jc in [lo, mid) (x), k in [mid, min(i, mid + KB - 1)], segment z
__global__ __launch_bounds__(WAVE) void k_cd_dc_w(CdGeom c, int lv, const double *__restrict__ T, double *W, double *Wt) {
    __shared__ double sA[CD_BLK], sB[CD_BLK];
    const int half = 1 << (lv - 1), lo = (int)blockIdx.z << lv, mid = lo + half, hi = min(lo + 2 * half, c.nblk);
    const int i = mid + (int)blockIdx.y, jc = lo + (int)blockIdx.x, lane = threadIdx.x;
    if (mid >= c.nblk || i >= hi) return;
    double acc[16];
#pragma unroll
    for (int x = 0; x < 16; x++) acc[x] = 0.0;
    cd_u4 pa[8], pb[8];
    const int k1 = min(i, min(mid + c.KB, hi) - 1);
    auto request = [&](int k) {
        cd_req_rows(Wt + (long)k * CD_NB * c.np + i * CD_NB, c.np, pa, lane);   // A = W_(i, k): [m][a] = the tile (k, i) of W^T, row-major
        cd_req_rows(T + (long)k * CD_NB * c.np + jc * CD_NB, c.np, pb, lane);
    };
    request(mid);
    for (int k = mid; k <= k1; k++) {
        cd_put(sA, pa, lane);
        cd_put(sB, pb, lane);
        if (k + 1 <= k1) request(k + 1);
        __syncthreads();
        cd_gemm_nt(acc, sA, sB, lane);
        __syncthreads();
    }
    const int tr = lane >> 3, tc = lane & 7;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        double *dst = W + (long)(i * CD_NB + 4 * tr + x) * c.np + jc * CD_NB + 4 * tc;
        *reinterpret_cast<cd_d2 *>(dst) = cd_d2{-acc[x * 4 + 0], -acc[x * 4 + 1]};
        *reinterpret_cast<cd_d2 *>(dst + 2) = cd_d2{-acc[x * 4 + 2], -acc[x * 4 + 3]};
    }
#pragma unroll
    for (int y = 0; y < 4; y++) {  // the transposed tile: row 4 tc + y of W^T's tile (jc, i), four consecutive entries 4 tr ..
        double *dst = Wt + (long)(jc * CD_NB + 4 * tc + y) * c.np + i * CD_NB + 4 * tr;
        *reinterpret_cast<cd_d2 *>(dst) = cd_d2{-acc[0 * 4 + y], -acc[1 * 4 + y]};
        *reinterpret_cast<cd_d2 *>(dst + 2) = cd_d2{-acc[2 * 4 + y], -acc[3 * 4 + y]};
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

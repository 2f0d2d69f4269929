// galerkin.h -- Galerkin coarse operators  A_c = P^T A P  (PCMGSetGalerkin(BOTH),
// LinearElasticity.cc:702) without ever forming a sparse matrix product.
//
// Every fine element lies inside exactly one coarse element and Q1
// interpolation inside a coarse element only involves that element's 8
// corners, therefore  P^T (sum_e K_e) P = sum_E K_E  with the coarse *element*
// matrix
//        K_E = sum_{8 children c}  (W_c (x) I3)^T  K_c  (W_c (x) I3),
// W_c[a][I] = trilinear weight of coarse corner I at node a of child c.  The
// Dirichlet part (I - N) is distributed over the elements sharing a node with
// weight 1/multiplicity (a power of two, exact), so that
//        A = sum_e ( E_e N_e KE N_e + D_e )            holds exactly.
// Level 0 -> 1 has a fast path K_E = sum_c E_c M_c (M_c = W_c^T KE W_c, 8
// constant matrices) for elements that touch no clamped node; elements that do
// are redone by a generic kernel.  Coarse element matrices are finally
// collapsed to the 27-point block stencil (DiaOp) used by the smoothers.
//
// Element matrices are stored element-major: Kel[elem * 576 + entry], entry = (3I+r)*24 + 3J+c:
// the kernels that build them work one coarse element per workgroup with the entries spread over
// the threads, so every load/store of a matrix is one contiguous 4.6 kB stream.
#pragma once
#include "common.h"
#include "operators.h"

// W[c][a][I], c = cx + 2cy + 4cz child position, a/I in reference corner order
__device__ __constant__ double c_W[512];

inline void host_W(double *W) {
    for (int c = 0; c < 8; c++) {
        const int cp[3] = {c & 1, (c >> 1) & 1, (c >> 2) & 1};
        for (int a = 0; a < 8; a++) {
            const int p[3] = {cp[0] + h_LX[a], cp[1] + h_LY[a], cp[2] + h_LZ[a]};  // 0..2, fine-node units
            for (int I = 0; I < 8; I++) {
                const int L[3] = {h_LX[I], h_LY[I], h_LZ[I]};
                double w = 1.0;
                for (int d = 0; d < 3; d++) w *= L[d] ? 0.5 * p[d] : 1.0 - 0.5 * p[d];
                W[(c * 8 + a) * 8 + I] = w;
            }
        }
    }
}

// M[c][576] = (W_c (x) I3)^T KE (W_c (x) I3)   (host, once per KE)
inline void host_child_matrices(const double *KE, double *M) {
    double W[512];
    host_W(W);
    for (int c = 0; c < 8; c++)
        for (int I = 0; I < 8; I++)
            for (int J = 0; J < 8; J++)
                for (int r = 0; r < 3; r++)
                    for (int cc = 0; cc < 3; cc++) {
                        double s = 0.0;
                        for (int a = 0; a < 8; a++)
                            for (int b = 0; b < 8; b++)
                                s += W[(c * 8 + a) * 8 + I] * KE[(3 * a + r) * 24 + 3 * b + cc] * W[(c * 8 + b) * 8 + J];
                        M[c * 576 + (3 * I + r) * 24 + 3 * J + cc] = s;
                    }
}

// M2[(c2*8 + g)][576] = (W_c2 (x) I3)^T M_g (W_c2 (x) I3): the level-2 element matrix is linear in the 64 fine moduli
// below it, K_E2 = sum_{c2, g} E_{c2 g} M2[c2][g]   (host, once per KE)
inline void host_grandchild_matrices(const double *M, double *M2) {
    double W[512];
    host_W(W);
    std::vector<double> T(576);
    for (int c2 = 0; c2 < 8; c2++)
        for (int g = 0; g < 8; g++) {
            const double *Mg = M + g * 576;
            // T = Mg (W_c2 (x) I3): columns contracted
            for (int ar = 0; ar < 24; ar++)
                for (int J = 0; J < 8; J++)
                    for (int cc = 0; cc < 3; cc++) {
                        double sacc = 0.0;
                        for (int b = 0; b < 8; b++) sacc += Mg[ar * 24 + 3 * b + cc] * W[(c2 * 8 + b) * 8 + J];
                        T[ar * 24 + 3 * J + cc] = sacc;
                    }
            double *out = M2 + (size_t)(c2 * 8 + g) * 576;
            for (int I = 0; I < 8; I++)
                for (int r = 0; r < 3; r++)
                    for (int col = 0; col < 24; col++) {
                        double sacc = 0.0;
                        for (int a = 0; a < 8; a++) sacc += W[(c2 * 8 + a) * 8 + I] * T[(3 * a + r) * 24 + col];
                        out[(3 * I + r) * 24 + col] = sacc;
                    }
        }
}

// ---- level 0 -> 2 in one step for level-2 elements without flagged children: thread = matrix entry (576 per
// workgroup), its 64 constants M2[.][entry] live in registers: 64 fma per entry, no intermediate level-1 matrices.
// The 64 moduli of an element are workgroup-uniform.  Round 3 read them with scalar loads: 128 SGPRs per element do not
// fit, the compiler split them into 8 groups with a wait behind each -- eight dependent scalar round trips per element,
// 551 us at 128^3 for 1.2 G fma (the kernel heads the longest chain of the set-up: -> level 3 -> level 4 -> factorisation).
// Round 4: the moduli of EIGHT elements are gathered by 512 threads (one value each), parked in registers while the
// previous batch is computed, and passed through a double-buffered LDS table that every thread reads as broadcasts.
// Same products; summed as four partial sums in a fixed order (round 3: one chain).
// A workgroup serves ONE THIRD of the 576 entries (192 threads = 3 waves; blockIdx.y = third): several small workgroups
// share a CU and hide each other's barrier and LDS latency, and a thread may use up to 256 registers.
constexpr int L2F_B = 6, L2F_T = 192;
__global__ __launch_bounds__(L2F_T, 2) void k_galerkin_l2_fast(Geom g0, Geom g2, const double *__restrict__ E,
                                                               const double *__restrict__ M2, double *__restrict__ Kc,
                                                               int nel) {
    __shared__ double sE[2][L2F_B][64];
    const int t = threadIdx.x + L2F_T * blockIdx.y;  // matrix entry
    double m2[64];
#pragma unroll
    for (int q = 0; q < 64; q++) m2[q] = M2[(size_t)q * 576 + t];
    const long sx = g0.ex, sxy = (long)g0.ex * g0.ey;  // row / layer pitch of the fine moduli
    // every own element is written here; the few with a flagged level-1 child are overwritten afterwards by the
    // generic construction (stream order), which spares a per-element flag load in this loop
    const int niter = ((int)blockIdx.x < nel) ? (nel - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;  // elements of this workgroup
    if (niter == 0) return;
    // loader role: thread -> two values of the batch (6 elements x 64 moduli = 384 = 2 x 192); modulus q = x + 4 y + 16 z
    int lb[2], lq[2];
    long qoff[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int f = (int)threadIdx.x + L2F_T * r;
        lb[r] = f >> 6;
        lq[r] = f & 63;
        qoff[r] = (lq[r] & 3) + sx * ((lq[r] >> 2) & 3) + sxy * (lq[r] >> 4);
    }
    auto fetch = [&](int i0, double v[2]) {  // (always valid addresses: beyond the end the last element is read again)
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int el = (int)blockIdx.x + min(i0 + lb[r], niter - 1) * (int)gridDim.x;
            const int I2 = el % g2.ex, J2 = (el / g2.ex) % g2.ey, K2 = el / (g2.ex * g2.ey);
            v[r] = E[(long)(4 * I2) + sx * (4 * J2) + sxy * (4 * K2) + qoff[r]];
        }
    };
    double pre[2];
    fetch(0, pre);
#pragma unroll
    for (int r = 0; r < 2; r++) sE[0][lb[r]][lq[r]] = pre[r];
    fetch(L2F_B, pre);
    __syncthreads();
    int cur = 0;
    for (int b0 = 0; b0 < niter; b0 += L2F_B, cur ^= 1) {
        const int nb = min(L2F_B, niter - b0);
        for (int j = 0; j < nb; j++) {
            // the 64 broadcast reads of an element in four groups of 16, group z + 1 requested before group z is consumed
            // (left to itself the compiler issues one read, waits, multiplies: 32 serial LDS round trips per element)
            typedef double l2f_d2 __attribute__((ext_vector_type(2)));
            const l2f_d2 *__restrict__ se = reinterpret_cast<const l2f_d2 *>(sE[cur][j]);
            l2f_d2 ev[2][8];
#pragma unroll
            for (int u = 0; u < 8; u++) ev[0][u] = se[u];
            // four partial sums (one per x position of the 4 x 4 x 4 block), added up at the end in a fixed order: a single
            // chain of 64 dependent FP64 fma left the two waves of a SIMD waiting on each other's latency
            double sa[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int z = 0; z < 4; z++) {
                if (z < 3) {
#pragma unroll
                    for (int u = 0; u < 8; u++) ev[(z + 1) & 1][u] = se[8 * (z + 1) + u];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int y = 0; y < 4; y++)
#pragma unroll
                    for (int x = 0; x < 4; x++) {
                        const int c2 = (x >> 1) + 2 * (y >> 1) + 4 * (z >> 1), g = (x & 1) + 2 * (y & 1) + 4 * (z & 1);
                        sa[x] = fma(ev[z & 1][(x + 4 * y) >> 1][x & 1], m2[c2 * 8 + g], sa[x]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            const double sacc = (sa[0] + sa[1]) + (sa[2] + sa[3]);
            Kc[(long)((int)blockIdx.x + (b0 + j) * (int)gridDim.x) * 576 + t] = sacc;
        }
#pragma unroll
        for (int r = 0; r < 2; r++) sE[cur ^ 1][lb[r]][lq[r]] = pre[r];   // batch b0 + B (in flight since the previous trip)
        fetch(b0 + 2 * L2F_B, pre);                                       // batch b0 + 2 B
        __syncthreads();
    }
}

// ---- level 0 -> 1, fast path: one 192-thread workgroup per own coarse element, 3 entries per thread
__global__ __launch_bounds__(192) void k_galerkin_fine_fast(Geom gf, Geom gc, const double *__restrict__ E,
                                                            const double *__restrict__ M, double *__restrict__ Kel) {
    const long t = blockIdx.x;  // own coarse element
    const int I = (int)(t % gc.ex), J = (int)((t / gc.ex) % gc.ey), K = (int)(t / ((long)gc.ex * gc.ey));
    double Ec[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int i = 2 * I + (c & 1), j = 2 * J + ((c >> 1) & 1), k = 2 * K + ((c >> 2) & 1);
        Ec[c] = E[(long)i + (long)gf.ex * (j + (long)gf.ey * k)];
    }
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int e = threadIdx.x + 192 * q;
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < 8; c++) s = fma(Ec[c], M[c * 576 + e], s);
        Kel[t * 576 + e] = s;
    }
}

// ---- level 0 -> 1, generic path for coarse elements touching clamped nodes --
// one 64-thread workgroup per listed coarse element, thread = (I, J) block
__global__ __launch_bounds__(64) void k_galerkin_fine_masked(Geom gf, Geom gc, const double *__restrict__ E,
                                                             const double *__restrict__ KE,
                                                             const uint8_t *__restrict__ mask,
                                                             const int *__restrict__ list, double *__restrict__ Kel,
                                                             int compact) {
    // interpolation weights and KE in LDS: both are indexed per lane (constant memory would serialise)
    __shared__ double s_KE[576], s_E[8];
    __shared__ unsigned s_m[27];
    for (int q = threadIdx.x; q < 576; q += 64) s_KE[q] = KE[q];
    const long t = list[blockIdx.x];
    const long slot = compact ? (long)blockIdx.x : t;  // compact: the matrix of the f-th listed element is row f
    const int Ie = (int)(t % gc.ex), Je = (int)((t / gc.ex) % gc.ey), Ke = (int)(t / ((long)gc.ex * gc.ey));
    // the element's 27 fine nodes' masks and 8 child moduli, ONE round trip up front (round 3 read a mask byte inside
    // every trip of the nested loops below, behind divergent `continue`s: ~600 dependent global loads per wave, 334 us for
    // the 4096 flagged elements of the 128^3 cantilever -- the head of the set-up's longest chain)
    if (threadIdx.x < 27) {
        const int di = threadIdx.x % 3, dj = (threadIdx.x / 3) % 3, dk = threadIdx.x / 9;
        s_m[threadIdx.x] = mask[(long)(2 * Ie + di) + (long)gf.nx * ((2 * Je + dj) + (long)gf.ny * (2 * Ke + dk))];
    } else if (threadIdx.x >= 32 && threadIdx.x < 40) {
        const int c = threadIdx.x - 32;
        const int i = 2 * Ie + (c & 1), j = 2 * Je + ((c >> 1) & 1), k = 2 * Ke + ((c >> 2) & 1);
        s_E[c] = E[(long)i + (long)gf.ex * (j + (long)gf.ey * k)];
    }
    __syncthreads();
    const int I = threadIdx.x >> 3, J = threadIdx.x & 7;
    double acc[9];
#pragma unroll
    for (int q = 0; q < 9; q++) acc[q] = 0.0;
    // The trilinear weight of coarse corner I at fine corner a of child c is a product over the three directions of
    // w1(c_d + a_d, I_d) in {1, 1/2, 0}.  Of the 8 combinations (c_d, a_d, b_d) of one direction only 4 or 5 give a non-zero
    // w1(c_d + a_d, I_d) w1(c_d + b_d, J_d): the lane walks ITS OWN 5 x 5 x 5 list (a zero-weight dummy pads the lists of 4)
    // instead of all 8 x 8 x 8 triples -- round 3 did, with 3/4 of the trips skipped by divergent `continue`s: the kernel
    // was bound by the instructions of trips that did nothing (277 us for 4096 elements).
    // per direction and list entry: c_d | a_d << 1 | b_d << 2, and w1(c_d + a_d, I_d) * w1(c_d + b_d, J_d); in LDS, a column
    // per lane (private arrays indexed by the loop counters would live in scratch memory)
    __shared__ int s_cab[15][64];
    __shared__ double s_wd[15][64];
    const int ln = threadIdx.x;
    {
        const int Ib[3] = {c_LX[I], c_LY[I], c_LZ[I]}, Jb[3] = {c_LX[J], c_LY[J], c_LZ[J]};
#pragma unroll
        for (int d = 0; d < 3; d++) {
            int n = 0;
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const int cd = t & 1, ad = (t >> 1) & 1, bd = (t >> 2) & 1;
                const int fa = cd + ad, fb = cd + bd;
                const double wI = fa == 1 ? 0.5 : (fa == 2 * Ib[d] ? 1.0 : 0.0), wJ = fb == 1 ? 0.5 : (fb == 2 * Jb[d] ? 1.0 : 0.0);
                if (wI * wJ != 0.0 && n < 5) {
                    s_cab[d * 5 + n][ln] = t;
                    s_wd[d * 5 + n][ln] = wI * wJ;
                    n++;
                }
            }
            for (; n < 5; n++) s_cab[d * 5 + n][ln] = 0, s_wd[d * 5 + n][ln] = 0.0;
        }
    }
    // corner number from its (x, y, z) bits (c_LX, c_LY, c_LZ: counter-clockwise in the lower plane, then the upper one)
    auto corner_of = [](int x, int y, int z) { return (y ? 3 - x : x) + 4 * z; };
    for (int tz = 0; tz < 5; tz++)
        for (int ty = 0; ty < 5; ty++)
            for (int tx = 0; tx < 5; tx++) {
                const double wIJ = s_wd[tx][ln] * s_wd[5 + ty][ln] * s_wd[10 + tz][ln];   // (powers of two: exact)
                if (wIJ == 0.0) continue;
                const int ex_ = s_cab[tx][ln], ey_ = s_cab[5 + ty][ln], ez_ = s_cab[10 + tz][ln];
                const int ci = ex_ & 1, cj = ey_ & 1, ck = ez_ & 1;      // child position inside the coarse element
                const int c = ci + 2 * cj + 4 * ck;
                const int ax = (ex_ >> 1) & 1, ay = (ey_ >> 1) & 1, az = (ez_ >> 1) & 1;
                const int bx = (ex_ >> 2) & 1, by = (ey_ >> 2) & 1, bz = (ez_ >> 2) & 1;
                const int a = corner_of(ax, ay, az), b = corner_of(bx, by, bz);
                const unsigned ma = s_m[(ci + ax) + 3 * ((cj + ay) + 3 * (ck + az))];
                const unsigned mb = s_m[(ci + bx) + 3 * ((cj + by) + 3 * (ck + bz))];
                const double w = wIJ * s_E[c];
                // a clamped row / column takes weight 0 (fma(0, k, acc) = acc: the same bits as skipping the term, no branch)
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int cc = 0; cc < 3; cc++) {
                        const double wm = (((ma >> r) | (mb >> cc)) & 1u) ? 0.0 : w;
                        acc[r * 3 + cc] = fma(wm, s_KE[(3 * a + r) * 24 + 3 * b + cc], acc[r * 3 + cc]);
                    }
                if (a == b && ma) {
                    // Dirichlet identity, shared between the elements around the node:
                    // multiplicity from the GLOBAL position of the node
                    const int ia = 2 * Ie + ci + ax, ja = 2 * Je + cj + ay, ka = 2 * Ke + ck + az;
                    const int kg = ka + gf.gz0;
                    const int mult = ((ia == 0 || ia == gf.nx - 1) ? 1 : 2) * ((ja == 0 || ja == gf.ny - 1) ? 1 : 2) *
                                     ((kg == 0 || kg == gf.nz_glob - 1) ? 1 : 2);
                    const double wi = wIJ / (double)mult;
                    for (int r = 0; r < 3; r++)
                        if ((ma >> r) & 1u) acc[r * 3 + r] += wi;
                }
            }
    for (int r = 0; r < 3; r++)
        for (int cc = 0; cc < 3; cc++) Kel[slot * 576 + (3 * I + r) * 24 + 3 * J + cc] = acc[r * 3 + cc];
}

// ---- level l -> l+1 (l >= 1): one wave per coarse element.
// K_E = sum_c W_c^T K_c W_c with W_c a tensor product of per-axis 2x2 weights
//   child on the low side : w[a][I] = {{1, 0}, {.5, .5}},  high side: {{.5, .5}, {0, 1}}
// -> six in-LDS contractions (3 row axes, 3 column axes) of the 576-entry child matrix instead of
// the dense 64x64 triple product (8x fewer flops, and every matrix is read exactly once, coalesced).
__device__ __constant__ int c_FLIP[3][8] = {{1, 0, 3, 2, 5, 4, 7, 6}, {3, 2, 1, 0, 7, 6, 5, 4}, {4, 5, 6, 7, 0, 1, 2, 3}};
// FROM_E (level 1 -> 2 when level 1 is applied matrix-free): the child matrices are never stored; an unflagged
// child is sum_g E_g M_g formed here from the fine moduli (same fma order as k_galerkin_fine_fast), a flagged one
// is row fidx[child] of the compact array Kf.
template <bool FROM_E>
__global__ __launch_bounds__(64) void k_galerkin_coarse(Geom gf, Geom gc, const double *__restrict__ Kf,
                                                        double *__restrict__ Kc, Geom g0, const double *__restrict__ E,
                                                        const double *__restrict__ M, const int *__restrict__ fidx,
                                                        long nel, const int *__restrict__ list) {
    __shared__ double A[576], B[576];
    const int t = threadIdx.x;
    // FROM_E: this thread's 9 entries of the 8 constant child matrices stay in registers for all elements of the block
    double Mr[FROM_E ? 9 : 1][FROM_E ? 8 : 1];
    if (FROM_E) {
#pragma unroll
        for (int q = 0; q < 9; q++)
#pragma unroll
            for (int g8 = 0; g8 < 8; g8++) Mr[q][g8] = M[g8 * 576 + t + 64 * q];
    }
    for (long idx = blockIdx.x; idx < nel; idx += gridDim.x) {  // own coarse elements (or the listed ones)
        const long el = list ? list[idx] : idx;
        const int Ie = (int)(el % gc.ex), Je = (int)((el / gc.ex) % gc.ey), Ke = (int)(el / ((long)gc.ex * gc.ey));
        double acc[9];
#pragma unroll
        for (int q = 0; q < 9; q++) acc[q] = 0.0;
        for (int c = 0; c < 8; c++) {
            const long ch = (long)(2 * Ie + (c & 1)) + (long)gf.ex * ((2 * Je + ((c >> 1) & 1)) + (long)gf.ey * (2 * Ke + ((c >> 2) & 1)));
            const int f = FROM_E ? fidx[ch] : 0;
            if (!FROM_E || f >= 0) {
                const double *__restrict__ src = Kf + (FROM_E ? (long)f : ch) * 576;
#pragma unroll
                for (int q = 0; q < 9; q++) A[t + 64 * q] = src[t + 64 * q];
            } else {
                const int i1 = (int)(ch % gf.ex), j1 = (int)((ch / gf.ex) % gf.ey), k1 = (int)(ch / ((long)gf.ex * gf.ey));
                double Eg[8];
#pragma unroll
                for (int g8 = 0; g8 < 8; g8++)
                    Eg[g8] = E[(long)(2 * i1 + (g8 & 1)) + (long)g0.ex * ((2 * j1 + ((g8 >> 1) & 1)) + (long)g0.ey * (2 * k1 + ((g8 >> 2) & 1)))];
#pragma unroll
                for (int q = 0; q < 9; q++) {
                    double sacc = 0.0;
#pragma unroll
                    for (int g8 = 0; g8 < 8; g8++) sacc = fma(Eg[g8], Mr[FROM_E ? q : 0][FROM_E ? g8 : 0], sacc);
                    A[t + 64 * q] = sacc;
                }
            }
            __syncthreads();
            double *in = A, *out = B;
            for (int pass = 0; pass < 6; pass++) {
                const int d = pass % 3;            // axis
                const bool rows = pass < 3;        // contract the row (a) or the column (b) node index
                const int hi = (c >> d) & 1;       // child on the high side of this axis
#pragma unroll
                for (int q = 0; q < 9; q++) {
                    const int o = t + 64 * q;
                    const int row = o / 24, col = o % 24;
                    const int nd = rows ? row / 3 : col / 3;  // node index being contracted (as coarse corner I_d)
                    const int lbit = d == 0 ? c_LX[nd] : (d == 1 ? c_LY[nd] : c_LZ[nd]);
                    const int fl = c_FLIP[d][nd];
                    const int o2 = rows ? (fl * 3 + row % 3) * 24 + col : row * 24 + fl * 3 + col % 3;
                    // in0 = value at fine node bit 0, in1 = at fine node bit 1 (other indices equal)
                    const double v_same = in[o], v_flip = in[o2];
                    const double in0 = lbit ? v_flip : v_same, in1 = lbit ? v_same : v_flip;
                    double r;
                    if (!hi) r = lbit ? 0.5 * in1 : in0 + 0.5 * in1;  // w = {{1,0},{.5,.5}}
                    else r = lbit ? 0.5 * in0 + in1 : 0.5 * in0;      // w = {{.5,.5},{0,1}}
                    out[o] = r;
                }
                __syncthreads();
                double *tmp = in;
                in = out;
                out = tmp;
            }
            // after 6 passes the result is back in A (in == A)
#pragma unroll
            for (int q = 0; q < 9; q++) acc[q] += in[t + 64 * q];
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < 9; q++) Kc[el * 576 + t + 64 * q] = acc[q];
    }
}

// ---- collapse element matrices to the 27-point block stencil ----------------
// thread = owned node, blockIdx.y = neighbour block (27).  Also writes the
// Jacobi inverse diagonal from the centre block.
__global__ __launch_bounds__(BLK) void k_elem_to_dia(Geom g, const double *__restrict__ Kel, double *__restrict__ S,
                                                     double *__restrict__ dinv) {
    const long nrows = 3 * g.nodes();
    const long plane = g.plane();
    const long t = blockIdx.x * (long)BLK + threadIdx.x;
    if (t >= g.owned_nodes()) return;
    const int blk = blockIdx.y;
    const int di = blk % 3 - 1, dj = (blk / 3) % 3 - 1, dk = blk / 9 - 1;
    const int k = g.own_lo + (int)(t / plane);
    const int rem = (int)(t % plane);
    const int j = rem / g.nx, i = rem % g.nx;
    const long n = t + plane * g.own_lo;
    double acc[9];
#pragma unroll
    for (int q = 0; q < 9; q++) acc[q] = 0.0;
    for (int I = 0; I < 8; I++) {
        // element in which this node is corner I
        const int ei = i - c_LX[I], ej = j - c_LY[I], ek = k - c_LZ[I];
        if (ei < 0 || ei >= g.ex || ej < 0 || ej >= g.ey || ek < 0 || ek >= g.ezl) continue;
        const int jx = c_LX[I] + di, jy = c_LY[I] + dj, jz = c_LZ[I] + dk;  // neighbour as corner of that element
        if (jx < 0 || jx > 1 || jy < 0 || jy > 1 || jz < 0 || jz > 1) continue;
        const int J = corner_of(jx, jy, jz);
        const long el = (long)ei + (long)g.ex * (ej + (long)g.ey * ek);
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int cc = 0; cc < 3; cc++) acc[r * 3 + cc] += Kel[el * 576 + (3 * I + r) * 24 + 3 * J + cc];
    }
#pragma unroll
    for (int cc = 0; cc < 3; cc++)
#pragma unroll
        for (int r = 0; r < 3; r++) S[(long)(blk * 3 + cc) * nrows + n * 3 + r] = acc[r * 3 + cc];
    if (blk == 13) {
#pragma unroll
        for (int r = 0; r < 3; r++) dinv[n * 3 + r] = 1.0 / acc[r * 3 + r];
    }
}

// Jacobi inverse diagonal straight from the coarse element matrices (levels whose
// operator is applied matrix-free): thread = owned node
__global__ __launch_bounds__(BLK) void k_elem_diag(Geom g, const double *__restrict__ Kel, double *__restrict__ dinv) {
    const long plane = g.plane();
    const long t = blockIdx.x * (long)BLK + threadIdx.x;
    if (t >= g.owned_nodes()) return;
    const int k = g.own_lo + (int)(t / plane);
    const int rem = (int)(t % plane);
    const int j = rem / g.nx, i = rem % g.nx;
    const long n = t + plane * g.own_lo;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int I = 0; I < 8; I++) {
        const int ei = i - c_LX[I], ej = j - c_LY[I], ek = k - c_LZ[I];
        if (ei < 0 || ei >= g.ex || ej < 0 || ej >= g.ey || ek < 0 || ek >= g.ezl) continue;
        const long el = (long)ei + (long)g.ex * (ej + (long)g.ey * ek);
#pragma unroll
        for (int r = 0; r < 3; r++) acc[r] += Kel[el * 576 + (3 * I + r) * 25];
    }
#pragma unroll
    for (int r = 0; r < 3; r++) dinv[n * 3 + r] = 1.0 / acc[r];
}

// Jacobi inverse diagonal of the matrix-free level 1 without its element matrices: unflagged elements contribute
// sum_c E_c diag(M_c) (same fma order as k_galerkin_fine_fast followed by k_elem_diag), flagged ones their compact row
__global__ __launch_bounds__(BLK) void k_macro_diag(Geom g0, Geom g, const double *__restrict__ E,
                                                    const double *__restrict__ M, const double *__restrict__ KelF,
                                                    const int *__restrict__ fidx, double *__restrict__ dinv) {
    const long plane = g.plane();
    const long t = blockIdx.x * (long)BLK + threadIdx.x;
    if (t >= g.owned_nodes()) return;
    const int k = g.own_lo + (int)(t / plane);
    const int rem = (int)(t % plane);
    const int j = rem / g.nx, i = rem % g.nx;
    const long n = t + plane * g.own_lo;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int I = 0; I < 8; I++) {
        const int ei = i - c_LX[I], ej = j - c_LY[I], ek = k - c_LZ[I];
        if (ei < 0 || ei >= g.ex || ej < 0 || ej >= g.ey || ek < 0 || ek >= g.ezl) continue;
        const long el = (long)ei + (long)g.ex * (ej + (long)g.ey * ek);
        const int f = fidx[el];
        if (f >= 0) {
#pragma unroll
            for (int r = 0; r < 3; r++) acc[r] += KelF[(long)f * 576 + (3 * I + r) * 25];
        } else {
            double s3[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const double Ec = E[(long)(2 * ei + (c & 1)) + (long)g0.ex * ((2 * ej + ((c >> 1) & 1)) + (long)g0.ey * (2 * ek + ((c >> 2) & 1)))];
#pragma unroll
                for (int r = 0; r < 3; r++) s3[r] = fma(Ec, M[c * 576 + (3 * I + r) * 25], s3[r]);
            }
#pragma unroll
            for (int r = 0; r < 3; r++) acc[r] += s3[r];
        }
    }
#pragma unroll
    for (int r = 0; r < 3; r++) dinv[n * 3 + r] = 1.0 / acc[r];
}

// ---- Dirichlet correction of the matrix-free level-1 operator -----------------
// dK[entry][f] = K_E (exact Galerkin element matrix, with N K N + D) - sum_c E_c M_c
// for the listed (flagged) coarse elements; thread = (entry, flagged element)
__global__ __launch_bounds__(BLK) void k_macro_delta(Geom gf, Geom gc, const double *__restrict__ E,
                                                     const double *__restrict__ M, const double *__restrict__ Kel,
                                                     const int *__restrict__ list, int nlist, double *__restrict__ dK) {
    const long t = blockIdx.x * (long)BLK + threadIdx.x;
    if (t >= (long)nlist * 576) return;
    const int f = (int)(t / 576), e = (int)(t % 576);  // e fastest: coalesced reads of Kel
    const long ce = list[f];
    const int I = (int)(ce % gc.ex), J = (int)((ce / gc.ex) % gc.ey), K = (int)(ce / ((long)gc.ex * gc.ey));
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int i = 2 * I + (c & 1), j = 2 * J + ((c >> 1) & 1), k = 2 * K + ((c >> 2) & 1);
        s = fma(E[(long)i + (long)gf.ex * (j + (long)gf.ey * k)], M[c * 576 + e], s);
    }
    dK[(long)e * nlist + f] = Kel[(long)f * 576 + e] - s;  // Kel: compact, row f = f-th listed element
}
// (A single-launch form -- one thread per (affected node, dof) running over its <= 8 x 24 entries -- measured
// 20.8 us against 9.8 + 5.5 us for the two launches below: too little parallelism per thread chain.)
// tmp[r][f] = dK_E[row r] . x_E ; thread = (row r of 24, flagged element f), coalesced over f
__global__ __launch_bounds__(BLK) void k_macro_corr_rows(Geom g, const double *__restrict__ dK,
                                                         const int *__restrict__ list, int nlist,
                                                         const double *__restrict__ x, double *__restrict__ tmp) {
    const long t = blockIdx.x * (long)BLK + threadIdx.x;
    if (t >= (long)nlist * 24) return;
    const int r = (int)(t / nlist), f = (int)(t % nlist);
    const long ce = list[f];
    const int ei = (int)(ce % g.ex), ej = (int)((ce / g.ex) % g.ey), ek = (int)(ce / ((long)g.ex * g.ey));
    double acc = 0.0;
#pragma unroll
    for (int J = 0; J < 8; J++) {
        const long nb = (long)(ei + LXc(J)) + (long)g.nx * ((ej + LYc(J)) + (long)g.ny * (ek + LZc(J)));
#pragma unroll
        for (int c = 0; c < 3; c++) acc = fma(dK[(long)(r * 24 + 3 * J + c) * nlist + f], x[3 * nb + c], acc);
    }
    tmp[t] = acc;
}
// corr[node] = sum over the flagged elements around it; adj[8*a + I] = flagged index of the
// element in which the node is corner I, or -1.  thread = affected node.
__global__ __launch_bounds__(BLK) void k_macro_corr_gather(const int *__restrict__ nodes, const int *__restrict__ adj,
                                                           int nnodes, const double *__restrict__ tmp,
                                                           double *__restrict__ corr, int nlist) {
    const int a = blockIdx.x * BLK + threadIdx.x;
    if (a >= nnodes) return;
    const long n = nodes[a];
    double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int I = 0; I < 8; I++) {
        const int f = adj[8 * a + I];
        if (f < 0) continue;
#pragma unroll
        for (int r = 0; r < 3; r++) acc[r] += tmp[(long)(3 * I + r) * nlist + f];
    }
#pragma unroll
    for (int r = 0; r < 3; r++) corr[3 * n + r] = acc[r];
}

// Gather of the element-row products around every affected node (as k_macro_corr_gather) applied directly to the
// result of the level-1 operator launch that computed them (matfree_tile.h: workgroups beyond the tiles):
//   APPLY  y += c (y += dinv c for the scaled product)      RESID  r -= c      CHEB  (d, x_out) -= c2 dinv c
template <int EPI>
__global__ __launch_bounds__(BLK) void k_macro_corr_apply(const int *__restrict__ nodes, const int *__restrict__ adj,
                                                          int nnodes, const double *__restrict__ tmp, int nlist,
                                                          NodeArgs a) {
    const int n0 = blockIdx.x * BLK + threadIdx.x;
    if (n0 >= nnodes) return;
    const long n = nodes[n0];
    double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int I = 0; I < 8; I++) {
        const int f = adj[8 * n0 + I];
        if (f < 0) continue;
#pragma unroll
        for (int r = 0; r < 3; r++) acc[r] += tmp[(long)(3 * I + r) * nlist + f];
    }
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const long q = 3 * n + r;
        if (EPI == EPI_APPLY) {
            a.out[q] += a.dinv ? acc[r] * a.dinv[q] : acc[r];  // a.dinv: the scaled product of the spectrum estimate
        } else if (EPI == EPI_RESID) {
            a.out[q] -= acc[r];
        } else if (EPI == EPI_CHEB) {
            const double f = a.c2 * (a.dinv[q] * acc[r]);
            a.d[q] -= f;
            a.out[q] -= f;
        }
    }
}

// matfree_tile.h -- the fine-level matrix-free hex8 operator, tuned for CDNA4.
//
//   y = (N K(E) N + I - N) u,   K = sum_e E_e KE          (LinearElasticity.cc:510-542)
//
// Why not the dense 24x24 product: 1152 flop per element against ~56 B of HBM
// traffic is FP64-issue bound at <50 % of the HBM roofline.  For
// a BOX element the stiffness commutes with the three reflections of the box,
// so in the Walsh-Hadamard basis over the 8 nodes (per displacement component)
// KE splits into eight 3x3 blocks:  KE = T^T B T,  T = H8 (x) I3  (B carries the
// 1/64).  That is 3*24 adds + 8*9 fma + 3*24 adds  (~240 FP64 ops, 4.8x fewer).
//
// Mapping: a 16x16 workgroup owns a 15x15 column of nodes and marches through a
// chunk of z-planes; one thread = one element column.
//   * nodal displacements of the incoming plane are staged through LDS (coalesced
//     row loads, Dirichlet mask applied while staging, double buffered, prefetched
//     one plane ahead in registers), the outgoing plane stays in registers;
//   * element results are combined to nodal sums WITHOUT atomics:
//       x: v_mov_dpp row_shr:1 inside the 16-lane row,  y: 6 doubles through LDS,
//       z: carried in registers to the next step;
//   * the epilogue (plain apply / residual / Chebyshev update / p.Ap) is fused.
// Redundancy: one halo element row/column per tile ((16/15)^2) and one extra
// layer per z-chunk.  Results are bitwise reproducible run to run.
#pragma once
#include "operators.h"

struct SymKE {
    double B[72];  // B[q*9 + r*3 + s], q = parity class, r/s = displacement component
};

// natural index m = lx + 2 ly + 4 lz  ->  reference corner number
static const int h_M2A[8] = {0, 1, 3, 2, 4, 5, 7, 6};

// B = blockdiag(T KE T^T) / 64.  Returns the largest |off-block entry| relative to
// the largest entry (0 for an exactly box-symmetric KE).
inline double make_sym_ke(const double *KE, SymKE *out) {
    double D[24][24];
    double maxabs = 0.0, maxoff = 0.0;
    for (int p = 0; p < 8; p++)
        for (int r = 0; r < 3; r++)
            for (int p2 = 0; p2 < 8; p2++)
                for (int s = 0; s < 3; s++) {
                    double acc = 0.0;
                    for (int m = 0; m < 8; m++)
                        for (int m2 = 0; m2 < 8; m2++) {
                            const int sg = (__builtin_popcount(p & m) + __builtin_popcount(p2 & m2)) & 1;
                            const double v = KE[(3 * h_M2A[m] + r) * 24 + 3 * h_M2A[m2] + s];
                            acc += sg ? -v : v;
                        }
                    D[p * 3 + r][p2 * 3 + s] = acc / 64.0;
                }
    for (int i = 0; i < 24; i++)
        for (int j = 0; j < 24; j++) {
            maxabs = fmax(maxabs, fabs(D[i][j]));
            const int qi = (i / 3) ^ (1 << (i % 3)), qj = (j / 3) ^ (1 << (j % 3));
            if (qi != qj) maxoff = fmax(maxoff, fabs(D[i][j]));
        }
    for (int q = 0; q < 8; q++)
        for (int r = 0; r < 3; r++)
            for (int s = 0; s < 3; s++) {
                // symmetrise the block (KE itself is symmetric only to rounding)
                const double a = D[(q ^ (1 << r)) * 3 + r][(q ^ (1 << s)) * 3 + s];
                const double b = D[(q ^ (1 << s)) * 3 + s][(q ^ (1 << r)) * 3 + r];
                out->B[q * 9 + r * 3 + s] = 0.5 * (a + b);
            }
    return maxabs > 0 ? maxoff / maxabs : 0.0;
}

__device__ inline double dpp_row_shr1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xF, 0xF, true);  // row_shr:1, zero fill
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// in-place 8-point Walsh-Hadamard butterfly on natural-order data
__device__ inline void wht8(double v[8]) {
#pragma unroll
    for (int h = 1; h < 8; h <<= 1)
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (!(i & h)) {
                const double a = v[i], b = v[i + h];
                v[i] = a + b;
                v[i + h] = a - b;
            }
}

// f = KE * u for one element; u, f indexed [natural node m][component]
__device__ inline void sym_ke_apply(const SymKE &S, double u[3][8], double f[3][8]) {
#pragma unroll
    for (int c = 0; c < 3; c++) wht8(u[c]);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const double vx = u[0][q ^ 1], vy = u[1][q ^ 2], vz = u[2][q ^ 4];
        f[0][q ^ 1] = fma(S.B[q * 9 + 0], vx, fma(S.B[q * 9 + 1], vy, S.B[q * 9 + 2] * vz));
        f[1][q ^ 2] = fma(S.B[q * 9 + 3], vx, fma(S.B[q * 9 + 4], vy, S.B[q * 9 + 5] * vz));
        f[2][q ^ 4] = fma(S.B[q * 9 + 6], vx, fma(S.B[q * 9 + 7], vy, S.B[q * 9 + 8] * vz));
    }
#pragma unroll
    for (int c = 0; c < 3; c++) wht8(f[c]);
}

constexpr int TILE = 16;             // threads per tile edge
constexpr int TOUT = TILE - 1;       // node columns produced per tile edge
constexpr int TSTG = TILE + 1;       // staged node columns per tile edge
constexpr int STG_N = TSTG * TSTG * 3;

template <int EPI>
__global__ __launch_bounds__(TILE * TILE) void k_matfree_tile(Geom g, const double *__restrict__ E,
                                                             const uint8_t *__restrict__ mask, SymKE S, NodeArgs a,
                                                             int KZ) {
    __shared__ double s_u[2][STG_N];
    __shared__ double s_y[TILE * TILE * 6];
    const int tid = threadIdx.x;
    const int tx = tid & (TILE - 1), ty = tid / TILE;
    const int bx = blockIdx.x * TOUT, by = blockIdx.y * TOUT;
    const int kz0 = g.own_lo + blockIdx.z * KZ;
    const int kz1 = min(kz0 + KZ - 1, g.own_hi);
    const int nsteps = kz1 - kz0 + 2;  // element layers kz0-1 .. kz1
    const int ei = bx - 1 + tx, ej = by - 1 + ty;
    const bool elem_ok = ei >= 0 && ei < g.ex && ej >= 0 && ej < g.ey;
    const bool node_ok = tx >= 1 && ty >= 1 && ei < g.nx && ej < g.ny;
    const long plane = g.plane();
    const double *__restrict__ x = a.x;

    // staging slots of this thread: flat index f -> (row, node column, component)
    int st_off[4];   // offset inside a node plane (doubles), -1 = outside the domain
    int st_node[4];  // node offset inside a plane
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int f = tid + s * TILE * TILE;
        st_off[s] = -1;
        st_node[s] = 0;
        if (f < STG_N) {
            const int r = f / (TSTG * 3), c = f % (TSTG * 3);
            const int gi = bx - 1 + c / 3, gj = by - 1 + r;
            if (gi >= 0 && gi < g.nx && gj >= 0 && gj < g.ny) {
                st_node[s] = gi + g.nx * gj;
                st_off[s] = 3 * st_node[s] + c % 3;
            }
        }
    }
    auto load_plane = [&](int p, double v[4]) {
        const bool pok = p >= 0 && p < g.nzl;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            double t = 0.0;
            if (pok && st_off[s] >= 0) {
                t = x[3 * plane * p + st_off[s]];
                if (mask && ((mask[plane * p + st_node[s]] >> (st_off[s] % 3)) & 1u)) t = 0.0;
            }
            v[s] = t;
        }
    };
    auto store_plane = [&](int buf, const double v[4]) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int f = tid + s * TILE * TILE;
            if (f < STG_N) s_u[buf][f] = v[s];
        }
    };
    // the 4 in-plane nodes of this thread's element, natural order (lx + 2 ly)
    const int o00 = (ty * TSTG + tx) * 3, o10 = o00 + 3, o01 = o00 + TSTG * 3, o11 = o01 + 3;
    auto read_nodes = [&](int buf, double u[3][8], int zoff) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            u[c][zoff + 0] = s_u[buf][o00 + c];
            u[c][zoff + 1] = s_u[buf][o10 + c];
            u[c][zoff + 2] = s_u[buf][o01 + c];
            u[c][zoff + 3] = s_u[buf][o11 + c];
        }
    };

    double pre[4];
    load_plane(kz0 - 1, pre);
    store_plane(0, pre);
    load_plane(kz0, pre);
    store_plane(1, pre);
    __syncthreads();
    double ubot[3][4];
    {
        double tmp[3][8];
        read_nodes(0, tmp, 0);
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int m = 0; m < 4; m++) ubot[c][m] = tmp[c][m];
    }
    double carry[3] = {0.0, 0.0, 0.0};
    double pdot = 0.0;

    for (int s = 0; s < nsteps; s++) {
        const int el = kz0 - 1 + s;  // element layer; bottom node plane el, top el+1
        double u[3][8], f[3][8];
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int m = 0; m < 4; m++) u[c][m] = ubot[c][m];
        read_nodes((s + 1) & 1, u, 4);
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int m = 0; m < 4; m++) ubot[c][m] = u[c][4 + m];
        const bool more = s + 1 < nsteps;
        if (more) load_plane(el + 2, pre);  // prefetch, consumed after the compute below

        double Ee = 0.0;
        if (elem_ok && el >= 0 && el < g.ezl) Ee = E[(long)ei + (long)g.ex * (ej + (long)g.ey * el)];
        sym_ke_apply(S, u, f);
        // nodal partial sums at this thread's node column: own element + left neighbour (DPP)
        double sB0[3], sB1[3], sT0[3], sT1[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double f0 = Ee * f[c][0], f1 = Ee * f[c][1], f2 = Ee * f[c][2], f3 = Ee * f[c][3];
            const double f4 = Ee * f[c][4], f5 = Ee * f[c][5], f6 = Ee * f[c][6], f7 = Ee * f[c][7];
            sB0[c] = f0 + dpp_row_shr1(f1);  // node (ei, ej  , el  )
            sB1[c] = f2 + dpp_row_shr1(f3);  // node (ei, ej+1, el  )
            sT0[c] = f4 + dpp_row_shr1(f5);  // node (ei, ej  , el+1)
            sT1[c] = f6 + dpp_row_shr1(f7);  // node (ei, ej+1, el+1)
        }
        // y-combination: pass the upper-row sums to the thread above
#pragma unroll
        for (int c = 0; c < 3; c++) {
            s_y[tid * 6 + c] = sB1[c];
            s_y[tid * 6 + 3 + c] = sT1[c];
        }
        __syncthreads();
        double yB[3], yT[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double b1 = ty >= 1 ? s_y[(tid - TILE) * 6 + c] : 0.0;
            const double t1 = ty >= 1 ? s_y[(tid - TILE) * 6 + 3 + c] : 0.0;
            yB[c] = carry[c] + (sB0[c] + b1);
            yT[c] = sT0[c] + t1;
        }
        if (s >= 1 && node_ok) {
            const long n = (long)ei + (long)g.nx * ej + plane * el;
            const unsigned m = mask ? mask[n] : 0u;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const long q = n * 3 + c;
                const double xq = x[q];
                const double y = ((m >> c) & 1u) ? xq : yB[c];
                if (EPI == EPI_APPLY) {
                    a.out[q] = y;
                } else if (EPI == EPI_RESID) {
                    a.out[q] = a.b[q] - y;
                } else if (EPI == EPI_CHEB) {
                    const double res = a.b[q] - y;
                    const double dn = a.c1 * a.d[q] + a.c2 * (a.dinv[q] * res);
                    a.d[q] = dn;
                    a.out[q] = xq + dn;
                } else {
                    a.out[q] = y;
                    pdot = fma(xq, y, pdot);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) carry[c] = yT[c];
        if (more) store_plane(s & 1, pre);
        __syncthreads();
    }
    if (EPI == EPI_APPLY_DOT) {
        pdot = block_sum(pdot);
        if (tid == 0) a.partials[blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)] = pdot;
    }
}

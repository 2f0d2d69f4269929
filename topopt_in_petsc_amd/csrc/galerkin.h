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
// Element matrices are stored entry-major ("SoA"): Kel[entry * nE + elem],
// entry = (3I+r)*24 + 3J+c, so that consecutive threads (elements) coalesce.
#pragma once
#include "common.h"

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

// ---- level 0 -> 1, fast path: one thread per own coarse element ------------
__global__ __launch_bounds__(BLK) void k_galerkin_fine_fast(Geom gf, Geom gc, const double *__restrict__ E,
                                                            const double *__restrict__ M, double *__restrict__ Kel) {
    const long nEc = gc.elems_stored();
    const long t = blockIdx.x * (long)BLK + threadIdx.x;
    if (t >= gc.own_elems()) return;
    const int I = (int)(t % gc.ex), J = (int)((t / gc.ex) % gc.ey), K = (int)(t / ((long)gc.ex * gc.ey));
    double Ec[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int i = 2 * I + (c & 1), j = 2 * J + ((c >> 1) & 1), k = 2 * K + ((c >> 2) & 1);
        Ec[c] = E[(long)i + (long)gf.ex * (j + (long)gf.ey * k)];
    }
    for (int e = 0; e < 576; e++) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < 8; c++) s = fma(Ec[c], M[c * 576 + e], s);
        Kel[(long)e * nEc + t] = s;
    }
}

// ---- level 0 -> 1, generic path for coarse elements touching clamped nodes --
// one 64-thread workgroup per listed coarse element, thread = (I, J) block
__global__ __launch_bounds__(64) void k_galerkin_fine_masked(Geom gf, Geom gc, const double *__restrict__ E,
                                                             const double *__restrict__ KE,
                                                             const uint8_t *__restrict__ mask,
                                                             const int *__restrict__ list, double *__restrict__ Kel) {
    const long nEc = gc.elems_stored();
    const long t = list[blockIdx.x];
    const int Ie = (int)(t % gc.ex), Je = (int)((t / gc.ex) % gc.ey), Ke = (int)(t / ((long)gc.ex * gc.ey));
    const int I = threadIdx.x >> 3, J = threadIdx.x & 7;
    double acc[9];
#pragma unroll
    for (int q = 0; q < 9; q++) acc[q] = 0.0;
    for (int c = 0; c < 8; c++) {
        const int i = 2 * Ie + (c & 1), j = 2 * Je + ((c >> 1) & 1), k = 2 * Ke + ((c >> 2) & 1);
        const double Ec = E[(long)i + (long)gf.ex * (j + (long)gf.ey * k)];
        for (int a = 0; a < 8; a++) {
            const double wa = c_W[(c * 8 + a) * 8 + I];
            if (wa == 0.0) continue;
            const int ia = i + c_LX[a], ja = j + c_LY[a], ka = k + c_LZ[a];
            const unsigned ma = mask[(long)ia + (long)gf.nx * (ja + (long)gf.ny * ka)];
            for (int b = 0; b < 8; b++) {
                const double wb = c_W[(c * 8 + b) * 8 + J];
                if (wb == 0.0) continue;
                const unsigned mb = mask[(long)(i + c_LX[b]) + (long)gf.nx * ((j + c_LY[b]) + (long)gf.ny * (k + c_LZ[b]))];
                const double w = wa * wb * Ec;
                for (int r = 0; r < 3; r++)
                    for (int cc = 0; cc < 3; cc++)
                        if (!((ma >> r) & 1u) && !((mb >> cc) & 1u))
                            acc[r * 3 + cc] = fma(w, KE[(3 * a + r) * 24 + 3 * b + cc], acc[r * 3 + cc]);
            }
            if (ma) {
                // Dirichlet identity, shared between the elements around the node:
                // multiplicity from the GLOBAL position of the node
                const int kg = ka + gf.gz0;
                const int mult = ((ia == 0 || ia == gf.nx - 1) ? 1 : 2) * ((ja == 0 || ja == gf.ny - 1) ? 1 : 2) *
                                 ((kg == 0 || kg == gf.nz_glob - 1) ? 1 : 2);
                const double w = wa * c_W[(c * 8 + a) * 8 + J] / (double)mult;
                for (int r = 0; r < 3; r++)
                    if ((ma >> r) & 1u) acc[r * 3 + r] += w;
            }
        }
    }
    for (int r = 0; r < 3; r++)
        for (int cc = 0; cc < 3; cc++) Kel[(long)((3 * I + r) * 24 + 3 * J + cc) * nEc + t] = acc[r * 3 + cc];
}

// ---- level l -> l+1 (l >= 1): thread = (coarse element, I, J) ---------------
__global__ __launch_bounds__(BLK) void k_galerkin_coarse(Geom gf, Geom gc, const double *__restrict__ Kf,
                                                         double *__restrict__ Kc) {
    const long nEf = gf.elems_stored(), nEc = gc.elems_stored(), nown = gc.own_elems();
    const long t = blockIdx.x * (long)BLK + threadIdx.x;
    if (t >= nown * 64) return;
    const int IJ = (int)(t / nown);
    const long el = t % nown;
    const int I = IJ >> 3, J = IJ & 7;
    const int Ie = (int)(el % gc.ex), Je = (int)((el / gc.ex) % gc.ey), Ke = (int)(el / ((long)gc.ex * gc.ey));
    double acc[9];
#pragma unroll
    for (int q = 0; q < 9; q++) acc[q] = 0.0;
    for (int c = 0; c < 8; c++) {
        const long ch = (long)(2 * Ie + (c & 1)) + (long)gf.ex * ((2 * Je + ((c >> 1) & 1)) + (long)gf.ey * (2 * Ke + ((c >> 2) & 1)));
        for (int a = 0; a < 8; a++) {
            const double wa = c_W[(c * 8 + a) * 8 + I];
            if (wa == 0.0) continue;
            for (int b = 0; b < 8; b++) {
                const double w = wa * c_W[(c * 8 + b) * 8 + J];
                if (w == 0.0) continue;
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int cc = 0; cc < 3; cc++)
                        acc[r * 3 + cc] = fma(w, Kf[(long)((3 * a + r) * 24 + 3 * b + cc) * nEf + ch], acc[r * 3 + cc]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 3; cc++) Kc[(long)((3 * I + r) * 24 + 3 * J + cc) * nEc + el] = acc[r * 3 + cc];
}

// ---- collapse element matrices to the 27-point block stencil ----------------
// thread = owned node, blockIdx.y = neighbour block (27).  Also writes the
// Jacobi inverse diagonal from the centre block.
__global__ __launch_bounds__(BLK) void k_elem_to_dia(Geom g, const double *__restrict__ Kel, double *__restrict__ S,
                                                     double *__restrict__ dinv) {
    const long nE = g.elems_stored();
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
            for (int cc = 0; cc < 3; cc++) acc[r * 3 + cc] += Kel[(long)((3 * I + r) * 24 + 3 * J + cc) * nE + el];
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
    const long nE = g.elems_stored();
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
        for (int r = 0; r < 3; r++) acc[r] += Kel[(long)((3 * I + r) * 25) * nE + el];
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
    const int e = (int)(t / nlist), f = (int)(t % nlist);
    const long ce = list[f];
    const int I = (int)(ce % gc.ex), J = (int)((ce / gc.ex) % gc.ey), K = (int)(ce / ((long)gc.ex * gc.ey));
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int i = 2 * I + (c & 1), j = 2 * J + ((c >> 1) & 1), k = 2 * K + ((c >> 2) & 1);
        s = fma(E[(long)i + (long)gf.ex * (j + (long)gf.ey * k)], M[c * 576 + e], s);
    }
    dK[t] = Kel[(long)e * gc.elems_stored() + ce] - s;
}
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

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
#include "macro_pattern.h"

// Non-zero pattern of the blocks (structural for any box element: q = 0 and 7 are
// full; a single-bit class only couples the two components other than that bit
// (the third is a rigid translation / pure shear pattern with zero energy);
// a two-bit class couples the two components of its bits and leaves the third
// one alone on the diagonal).  33 structural values -> they live in SGPRs.
//
// Round 6: + 3.  The (b, b) entry of the single-bit class q = 1 << b is the element's answer to a rigid translation
// along b, D[(0,b),(0,b)] = sum of the 64 entries KE[(m,b),(m2,b)] / 64: exactly 0 for an exact box element, ~1e-16 max|KE|
// for the reference's KE (LinearElasticity.cc:841-998, a Gauss sum in double whose rows do not quite sum to 0).  The
// translation mode of an element has amplitude ~1e3 against strains ~1e-2, and in uniform-modulus regions this mean residue
// is the only part of KE's rounding residue that survives assembly: dropping it moved the compliance of the 128^3 cantilever
// by 1.59e-10 (1e-10 is the contract); with the three values kept the packed form follows KE to 5e-13 (DESIGN 2.1;
// tools/r06_ke_residue.py ranks all 576 entries of the residue by their share).  -DSYMKE_NO_TRANSLATION_RESIDUE: rounds 1-5.
#ifdef SYMKE_NO_TRANSLATION_RESIDUE
constexpr bool SYMKE_TRANSL = false;
#else
constexpr bool SYMKE_TRANSL = true;
#endif
__host__ __device__ constexpr bool symke_nz(int q, int r, int s) {
    if (q == 0 || q == 7) return true;
    if (q == 1 || q == 2 || q == 4) {
        const int b = q == 1 ? 0 : (q == 2 ? 1 : 2);
        return (r != b && s != b) || (SYMKE_TRANSL && r == b && s == b);
    }
    const int m = q == 6 ? 0 : (q == 5 ? 1 : 2);  // the component whose bit is NOT in q
    return (r != m && s != m) || (r == m && s == m);
}
// index of the (symmetric) value B_q[r][s] in the packed array, -1 if structurally zero
__host__ __device__ constexpr int symke_idx(int q, int r, int s) {
    if (!symke_nz(q, r, s)) return -1;
    const int lo = r < s ? r : s, hi = r < s ? s : r;
    int n = 0;
    for (int qq = 0; qq < 8; qq++)
        for (int a = 0; a < 3; a++)
            for (int b = a; b < 3; b++) {
                if (!symke_nz(qq, a, b)) continue;
                if (qq == q && a == lo && b == hi) return n;
                n++;
            }
    return -1;
}
constexpr int SYMKE_N = SYMKE_TRANSL ? 36 : 33;
static_assert(symke_idx(7, 2, 2) == SYMKE_N - 1, "packed size");

constexpr int SYMKE_NTOT = SYMKE_N + 3;  // + 1 / KE[c][c]: the nodal diagonal is KE[c][c] * (sum of the 8 adjacent moduli)
// Round 6, the KRYLOV operator's extra entries.  The operator of the Krylov method itself (EPI_APPLY_DOT: A p of CG, and the
// initial residual A x0; tp_elasticity_apply_krylov) also applies what is left of T KE T / 64 in the translation mode's
// COLUMN (how every mode of the element answers a rigid translation: D[(p,r),(0,s)], 69 entries beside the three of the packed
// form) and ROW (how the translation mode answers every other mode: D[(0,r),(p,s)], p >= 1, 63 entries) -- one-sided, as KE has
// them, not averaged: KE's asymmetry (1.4e-17) is 10 % of these residues.  Why: a late CG residual is ~1e-5 ||b|| while the
// iterate's translation is ~1e5 x its strain, so KE's O(1e-16) answer to that translation is worth 1e-9 of ||r_k|| (C2, 35
// iterations; 1.6e-10 at C3) -- and the 80-bit arbiter shows that ONLY the Krylov operator carries this: with the Krylov products
// from KE and the whole preconditioner from the packed form, the reference's residual history is reproduced to 7e-15; with
// these 132 entries to 8e-13 (tools/r06_arbiter_sets.py).  The smoother and the residual inside the V-cycle keep the 36-value form
// (4 of the 5 fine-level operator applications of a Krylov iteration), and so does the plain product EPI_APPLY (MatMult, the
// spectrum estimates): +132 fma on ~170 FP64 instructions cost the product 36.6 -> 59.5 us at 128^3 (measured; 12.21 -> 12.40 ms per
// design iteration, 1.6 %).  -DSYMKE_NO_KRYLOV_RESIDUE: off.
#ifdef SYMKE_NO_KRYLOV_RESIDUE
constexpr bool SYMKE_KRYLOV = false;
#else
constexpr bool SYMKE_KRYLOV = SYMKE_TRANSL;
#endif
constexpr int SYMKE_XCOL = 72, SYMKE_XROW = 63, SYMKE_XN = SYMKE_XCOL + SYMKE_XROW;
struct SymKE {
    double a[SYMKE_NTOT];
    double x[SYMKE_XN];  // column [(p * 3 + r) * 3 + s] (0 where the packed form holds the entry), row [72 + (r * 7 + p - 1) * 3 + s]
};
__host__ __device__ constexpr bool symx_col(int p, int r, int s) { return !(p == 0 && r == s); }

// natural index m = lx + 2 ly + 4 lz  ->  reference corner number
static const int h_M2A[8] = {0, 1, 3, 2, 4, 5, 7, 6};

// Packs blockdiag(T KE T^T) / 64.  Returns the largest entry that the packed form
// drops (off-block or structurally-zero position), relative to the largest entry:
// ~1e-17 for a box element, O(1) if KE is not box symmetric.
inline double make_sym_ke(const double *KE, SymKE *out) {
    static double D[24][24];
    double maxabs = 0.0, dropped = 0.0;
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
            if (qi != qj || !symke_nz(qi, i % 3, j % 3)) dropped = fmax(dropped, fabs(D[i][j]));
        }
    for (int q = 0; q < 8; q++)
        for (int r = 0; r < 3; r++)
            for (int s = r; s < 3; s++) {
                const int id = symke_idx(q, r, s);
                if (id < 0) continue;
                const double a = D[(q ^ (1 << r)) * 3 + r][(q ^ (1 << s)) * 3 + s];
                const double b = D[(q ^ (1 << s)) * 3 + s][(q ^ (1 << r)) * 3 + r];
                out->a[id] = 0.5 * (a + b);  // KE itself is symmetric only to rounding
            }
    if (SYMKE_TRANSL)
        for (int r = 0; r < 3; r++) {
            // 64 entries that cancel to ~1e-16 of their size: summed in 80-bit arithmetic (exact here), rounded once --
            // the double-precision sum above carries a rounding error of 10-20 % of the value (the host restatement the
            // parity checks use does the same)
            long double acc = 0.0L;
            for (int m = 0; m < 8; m++)
                for (int m2 = 0; m2 < 8; m2++) acc += (long double)KE[(3 * h_M2A[m] + r) * 24 + 3 * h_M2A[m2] + r];
            out->a[symke_idx(1 << r, r, r)] = (double)(acc / 64.0L);
        }
    {   // the translation column and row, each entry the exact (80-bit) sum of its 64 terms rounded once
        auto Dx = [&](int p, int r, int p2, int s2) -> double {
            long double acc = 0.0L;
            for (int m = 0; m < 8; m++)
                for (int m2 = 0; m2 < 8; m2++) {
                    const int sg = (__builtin_popcount(p & m) + __builtin_popcount(p2 & m2)) & 1;
                    const long double v = (long double)KE[(3 * h_M2A[m] + r) * 24 + 3 * h_M2A[m2] + s2];
                    acc += sg ? -v : v;
                }
            return (double)(acc / 64.0L);
        };
        for (int p = 0; p < 8; p++)
            for (int r = 0; r < 3; r++)
                for (int s2 = 0; s2 < 3; s2++) out->x[(p * 3 + r) * 3 + s2] = symx_col(p, r, s2) ? Dx(p, r, 0, s2) : 0.0;
        for (int r = 0; r < 3; r++)
            for (int p = 1; p < 8; p++)
                for (int s2 = 0; s2 < 3; s2++) out->x[SYMKE_XCOL + (r * 7 + p - 1) * 3 + s2] = Dx(0, r, p, s2);
    }
    for (int c = 0; c < 3; c++) {
        out->a[SYMKE_N + c] = 1.0 / KE[c * 24 + c];
        for (int m = 1; m < 8; m++)  // equal at all 8 corners for a box element
            dropped = fmax(dropped, fabs(KE[(3 * m + c) * 24 + 3 * m + c] - KE[c * 24 + c]));
    }
    return maxabs > 0 ? dropped / maxabs : 0.0;
}

__device__ inline double dpp_row_shr1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xF, 0xF, true);  // row_shr:1, zero fill
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// 2-D Walsh-Hadamard butterfly over the 4 in-plane nodes (index lx + 2 ly); self-inverse up to 1/4
__device__ inline void wht4(double v[4]) {
    const double a = v[0] + v[1], b = v[0] - v[1], c = v[2] + v[3], d = v[2] - v[3];
    v[0] = a + c;
    v[1] = b + d;
    v[2] = a - c;
    v[3] = b - d;
}

// The packed blocks live in constant memory so that they are fetched with SCALAR
// loads (s_load -> SGPR operands of v_fma_f64: no vector registers, no LDS).  A
// small pool of slots lets several solver contexts (different KE) coexist.
constexpr int SYMKE_SLOTS = 16, SYMKE_STRIDE = 40;
static_assert(SYMKE_N + 3 <= SYMKE_STRIDE, "stride");
__constant__ double c_symB[SYMKE_SLOTS * SYMKE_STRIDE];
constexpr int SYMKE_XSTRIDE = 4 * SYMKE_STRIDE;  // the Krylov operator's extra entries of slot i start at 4 * (offset of slot i in c_symB)
static_assert(SYMKE_XN <= SYMKE_XSTRIDE, "stride");
__constant__ double c_symX[SYMKE_SLOTS * SYMKE_XSTRIDE];

// level-1 operator constants (MACG_N packed values of G_sigma, macro_pattern.h), same slot numbering
constexpr int MACG_STRIDE = 80;
static_assert(MACG_N <= MACG_STRIDE, "stride");
__constant__ double c_macG[SYMKE_SLOTS * MACG_STRIDE];

// G_sigma = H (1/8 sum_c chi_sigma(c) M_c) H^T / 64 at the pattern positions; M = the 8 child matrices W_c^T KE W_c
// (galerkin.h: host_child_matrices, reference corner numbering).  Returns the largest dropped entry (relative).
inline double make_macro_tensor(const double *M, double *out) {
    static double G[8][24][24];
    double maxabs = 0.0;
    for (int sg = 0; sg < 8; sg++)
        for (int p = 0; p < 8; p++)
            for (int r = 0; r < 3; r++)
                for (int p2 = 0; p2 < 8; p2++)
                    for (int s = 0; s < 3; s++) {
                        double acc = 0.0;
                        for (int c = 0; c < 8; c++) {
                            double a2 = 0.0;
                            for (int m = 0; m < 8; m++)
                                for (int m2 = 0; m2 < 8; m2++) {
                                    const int sgn = (__builtin_popcount(p & m) + __builtin_popcount(p2 & m2)) & 1;
                                    const double v = M[c * 576 + (3 * h_M2A[m] + r) * 24 + 3 * h_M2A[m2] + s];
                                    a2 += sgn ? -v : v;
                                }
                            acc += (__builtin_popcount(sg & c) & 1) ? -a2 : a2;
                        }
                        G[sg][3 * p + r][3 * p2 + s] = acc / 512.0;
                        maxabs = fmax(maxabs, fabs(acc / 512.0));
                    }
    double asym = 0.0;
    for (int c = 0; c < MACG_N; c++) {  // class representatives
        const int k = MACG_REP[c];
        out[c] = G[MACG_M_SIG[k]][MACG_M_ROW[k]][MACG_M_COL[k]];
    }
    for (int k = 0; k < MACG_M; k++) {  // every listed entry (and its mirror image) must be +-its class constant
        double &g = G[MACG_M_SIG[k]][MACG_M_ROW[k]][MACG_M_COL[k]], &gt = G[MACG_M_SIG[k]][MACG_M_COL[k]][MACG_M_ROW[k]];
        const double ref = MACG_M_SGN[k] * out[MACG_M_CLS[k]];
        asym = fmax(asym, fmax(fabs(g - ref), fabs(gt - ref)));
        g = gt = 0.0;
    }
    double dropped = 0.0;
    for (int sg = 0; sg < 8; sg++)
        for (int i = 0; i < 24; i++)
            for (int j = 0; j < 24; j++) dropped = fmax(dropped, fabs(G[sg][i][j]));
    dropped = fmax(dropped, asym);
    return maxabs > 0 ? dropped / maxabs : 0.0;
}
inline int macro_slot_upload(int slot, const double *vals) {
    return hipMemcpyToSymbol(HIP_SYMBOL(c_macG), vals, sizeof(double) * MACG_N, sizeof(double) * slot * MACG_STRIDE) ==
                   hipSuccess ? 0 : -1;
}

struct SymSlots {
    SymKE key[SYMKE_SLOTS];
    int refs[SYMKE_SLOTS];
};
inline SymSlots &sym_slots() {
    static SymSlots s = {};
    return s;
}
// returns a slot holding `sk` (shared between equal matrices), or -1 if the pool is exhausted
inline int sym_slot_acquire(const SymKE &sk) {
    SymSlots &S = sym_slots();
    for (int i = 0; i < SYMKE_SLOTS; i++)
        if (S.refs[i] > 0 && std::memcmp(&S.key[i], &sk, sizeof(SymKE)) == 0) {
            S.refs[i]++;
            return i;
        }
    for (int i = 0; i < SYMKE_SLOTS; i++)
        if (S.refs[i] == 0) {
            if (hipMemcpyToSymbol(HIP_SYMBOL(c_symB), sk.a, sizeof(sk.a), sizeof(double) * i * SYMKE_STRIDE) != hipSuccess)
                return -1;
            if (hipMemcpyToSymbol(HIP_SYMBOL(c_symX), sk.x, sizeof(sk.x), sizeof(double) * i * SYMKE_XSTRIDE) != hipSuccess)
                return -1;
            std::memcpy(&S.key[i], &sk, sizeof(SymKE));
            S.refs[i] = 1;
            return i;
        }
    return -1;
}
inline void sym_slot_release(int i) {
    if (i >= 0 && sym_slots().refs[i] > 0) sym_slots().refs[i]--;
}

// fhat = B uhat for one element in the Walsh-Hadamard basis; [component][p], p = px + 2 py + 4 pz.
__device__ inline void sym_ke_blocks(const double *B, const double u[3][8], double f[3][8]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const double v[3] = {u[0][q ^ 1], u[1][q ^ 2], u[2][q ^ 4]};
#pragma unroll
        for (int r = 0; r < 3; r++) {
            double acc = 0.0;
            bool first = true;
#pragma unroll
            for (int s = 0; s < 3; s++) {
                const int id = symke_idx(q, r, s);
                if (id >= 0) {
                    acc = first ? B[id] * v[s] : fma(B[id], v[s], acc);
                    first = false;
                }
            }
            f[r][q ^ (1 << r)] = acc;
        }
    }
}

// The Krylov operator's share (see SYMKE_KRYLOV): fhat += column . (translation part of uhat) and row . (the other modes).
// X: the slot's entries in c_symX (scalar loads).  The three row sums are chains of 21 fma each, split in two.
template <bool ON>
__device__ inline void sym_ke_translation(const double *X, const double u[3][8], double f[3][8]) {
#pragma clang fp contract(off)
    if (!ON) return;
#ifdef SYMKE_X_GROUPS
    // experiment: the scalar loads of the 135 constants in groups of 9 behind scheduling barriers (the compiler otherwise
    // hoists them all and spills SGPRs into vector lanes)
#define SYMX_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SYMX_FENCE() ((void)0)
#endif
#pragma unroll
    for (int p = 0; p < 8; p++) {
        SYMX_FENCE();
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int s = 0; s < 3; s++)
                if (symx_col(p, r, s)) f[r][p] = fma(X[(p * 3 + r) * 3 + s], u[s][0], f[r][p]);
    }
    double a0[3] = {0.0, 0.0, 0.0}, a1[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int p = 1; p < 8; p++) {
        SYMX_FENCE();
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int s = 0; s < 3; s++) {
                const double c = X[SYMKE_XCOL + (r * 7 + p - 1) * 3 + s];
                if (p & 1) a0[r] = fma(c, u[s][p], a0[r]);
                else a1[r] = fma(c, u[s][p], a1[r]);
            }
    }
    SYMX_FENCE();
#pragma unroll
    for (int r = 0; r < 3; r++) f[r][0] = f[r][0] + (a0[r] + a1[r]);
#undef SYMX_FENCE
}
template <int EPI>
struct KrylovEpi {   // the product of the Krylov method (A p with its p . A p; the initial residual goes through it too)
    static constexpr bool value = SYMKE_KRYLOV && EPI == EPI_APPLY_DOT;
};

constexpr int TILE = 16;             // threads per tile edge
constexpr int TOUT = TILE - 1;       // node columns produced per tile edge
constexpr int TSTG = TILE + 1;       // staged node columns per tile edge
constexpr int STG_N = TSTG * TSTG * 3;

// per-launch constants that do not depend on the epilogue
struct TileArgs {
    int nx, ny, nzl, ex, ey, ezl, own_lo, own_hi, kz;
    const double *E;           // [stored elements]
    const uint8_t *mask;       // per node clamped-dof bits (may be null)
    const uint8_t *colmask;    // per node COLUMN: OR of mask over the planes (may be null)
    int slot_off;              // offset of the packed SymKE inside c_symB
    // MACRO (level-1 Galerkin operator applied from the fine densities):
    int fex, fey;              // FINE element counts per row / column (children indexing)
    const double *corr;        // [level-1 dofs] Dirichlet correction added to y (k_macro_corr), or null
    int xcd_remap;             // 1: contiguous tile ranges per XCD
    int macg_off;              // offset of the level-1 constants inside c_macG
    int r1_lo, r1_hi;          // second range of output planes (empty if r1_lo > r1_hi): the launch that produces the
                               // two boundary planes of a slab first (halo overlap); chunks of the first range come first
    // MACRO, fused Dirichlet correction: workgroups beyond `ntiles` (0 = all are tiles) compute the element-row
    // products dK_E[row] . x_E of the flagged elements (k_macro_corr_rows) INSIDE the operator's launch -- they fill
    // the tail of the tile workgroups instead of a launch of their own; k_macro_corr_apply adds them afterwards
    int ntiles;
    const double *cr_dK;
    const int *cr_list;
    int cr_nflag;
    double *cr_tmp;
};
// output planes [kz0, kz1] of z-chunk bzi
__device__ inline void tile_chunk(const TileArgs &t, int bzi, int &kz0, int &kz1) {
    const int na = (t.own_hi - t.own_lo + t.kz) / t.kz;  // chunks of the first range (0 if empty)
    if (bzi < na) {
        kz0 = t.own_lo + bzi * t.kz;
        kz1 = min(kz0 + t.kz - 1, t.own_hi);
    } else {
        kz0 = t.r1_lo + (bzi - na) * t.kz;
        kz1 = min(kz0 + t.kz - 1, t.r1_hi);
    }
}

// MACRO = 0: fine level, one element per thread and step.
// MACRO = 1: level 1.  The Galerkin operator P^T A_0 P is never stored: a coarse element's matrix is linear in its 8
//   child moduli, K_E = sum_c E_c W_c^T KE W_c, and the child matrices are reflections of each other.  In the
//   Walsh-Hadamard basis of the 8 corners AND of the 8 children
//       y = T^T [ sum_sigma ehat_sigma G_sigma ] T u,      ehat = H8 E_children,
//   with constant SYMMETRIC G_sigma that couple mode class q only to q ^ sigma: 279 structurally non-zero entries,
//   150 in the upper triangles, 78 distinct constants for any box (macro_pattern.h, generated: entries that are
//   equal up to sign share one scalar load; per entry one scaling by ehat and two fma) instead of 8 dense child products -> ~430 FP64 ops per coarse element.  Bytes per
//   apply drop from 1944 B per coarse node (stored stencil) to the 8 child densities.
//   This kernel applies the operator WITHOUT Dirichlet conditions; the (few) coarse
//   elements that contain a clamped fine node differ from it by a stored 24x24
//   matrix dK_E whose action k_macro_corr precomputes into `corr`.
template <int EPI, int MACRO>
__global__ __launch_bounds__(TILE * TILE, MACRO ? 2 : 3) void k_matfree_tile(TileArgs t, NodeArgs a) {
    __shared__ double s_u[3][4 * TILE * TILE];  // node-plane ring: bottom, top, next (STG_N used, padded: unconditional stores)
    __shared__ double s_y[2][TILE * TILE * 3];  // y-combination, double buffered -> one barrier per step
    // fine-level CHEB streams 4 node vectors instead of 6: the Jacobi diagonal is rebuilt from the moduli (KE[c][c] * sum
    // of the 8 adjacent E, combined in x/y/z like the operator itself), and the recurrence runs in its 3-term form
    // u+ = u + c1 (u - u-) + c2 D^-1 (b - K u) with u- read from, and u+ written to, the same output slot
    constexpr bool DIAG_FLY = (EPI == EPI_CHEB && !MACRO);
    __shared__ double s_e[DIAG_FLY ? 2 : 1][DIAG_FLY ? TILE * TILE : 1];
    const int tid = threadIdx.x;
    const int tx = tid & (TILE - 1), ty = tid / TILE;
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (private L2 each); give every XCD a
    // contiguous run of tiles so that neighbouring tiles, which share halo columns, meet in one L2
    int bxi, byi, bzi;
    {
        const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int nb = t.ntiles > 0 ? t.ntiles : gridDim.x * gridDim.y * gridDim.z;
        if (MACRO && lin >= nb) {  // workgroup-uniform, before any barrier
            const long tt = (long)(lin - nb) * (TILE * TILE) + tid;
            if (tt < (long)t.cr_nflag * 24) {
                const int r = (int)(tt / t.cr_nflag), f = (int)(tt % t.cr_nflag);
                const long ce = t.cr_list[f];
                const int ci = (int)(ce % t.ex), cj = (int)((ce / t.ex) % t.ey), ck = (int)(ce / ((long)t.ex * t.ey));
                double acc = 0.0;
#pragma unroll
                for (int J = 0; J < 8; J++) {
                    const long nbn = (long)(ci + LXc(J)) + (long)t.nx * ((cj + LYc(J)) + (long)t.ny * (ck + LZc(J)));
#pragma unroll
                    for (int c = 0; c < 3; c++) acc = fma(t.cr_dK[(long)(r * 24 + 3 * J + c) * t.cr_nflag + f], a.x[3 * nbn + c], acc);
                }
                t.cr_tmp[tt] = acc;
            }
            return;
        }
        const int x8 = lin & 7;
        const int m = t.xcd_remap ? x8 * (nb >> 3) + min(x8, nb & 7) + (lin >> 3) : lin;  // bijection of [0, nb)
        bxi = m % gridDim.x;
        byi = (m / gridDim.x) % gridDim.y;
        bzi = m / (gridDim.x * gridDim.y);
    }
    const int bx = bxi * TOUT, by = byi * TOUT;
    int kz0, kz1;
    tile_chunk(t, bzi, kz0, kz1);
    const int nsteps = kz1 - kz0 + 2;  // element layers kz0-1 .. kz1
    const int ei = bx - 1 + tx, ej = by - 1 + ty;
    const bool elem_ok = ei >= 0 && ei < t.ex && ej >= 0 && ej < t.ey;
    const bool node_ok = tx >= 1 && ty >= 1 && ei < t.nx && ej < t.ny;
    const long plane = (long)t.nx * t.ny;
    const double *__restrict__ x = a.x;
    const unsigned eoff = elem_ok ? (unsigned)(ei + t.ex * ej) : 0u;  // always a valid element column
    const double emul = elem_ok ? 1.0 : 0.0;

    // ---- staging slots of this thread: flat index f -> (row, node column, component).
    // Branch-free hot path: every slot holds a VALID offset (columns outside the domain are clamped onto the
    // boundary column; their values only reach elements outside the domain, whose modulus is 0), the fourth slot
    // of the threads beyond STG_N duplicates the last entry into LDS padding, and the Dirichlet mask is only
    // consulted in tiles that contain a clamped column at all (workgroup-uniform flag).
    unsigned st_off[4];   // offset inside a node plane (doubles)
    unsigned st_cm = 0;   // bit s set: slot s belongs to a column with a clamped dof somewhere
    const long ncol = node_ok ? (long)ei + (long)t.nx * ej : 0;   // own output node column
    const unsigned own_cm = (node_ok && t.colmask) ? t.colmask[ncol] : 0u;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int f = min(tid + s * TILE * TILE, STG_N - 1);
        const int r = f / (TSTG * 3), c = f % (TSTG * 3);
        const int gi0 = bx - 1 + c / 3, gj0 = by - 1 + r;
        const int gi = min(max(gi0, 0), t.nx - 1), gj = min(max(gj0, 0), t.ny - 1);
        st_off[s] = 3u * (unsigned)(gi + t.nx * gj) + (unsigned)(c % 3);
        if (t.colmask && gi == gi0 && gj == gj0 && ((t.colmask[gi + t.nx * gj] >> (c % 3)) & 1u)) st_cm |= 1u << s;
    }
    const bool tile_masked = t.colmask ? (__syncthreads_or((st_cm | own_cm) != 0u) != 0) : false;
    auto load_plane = [&](int p, double v[4]) {
        if (p < 0 || p >= t.nzl) {  // uniform
#pragma unroll
            for (int s = 0; s < 4; s++) v[s] = 0.0;
            return;
        }
        const double *__restrict__ xp = x + 3 * plane * p;
#pragma unroll
        for (int s = 0; s < 4; s++) v[s] = xp[st_off[s]];
        if (tile_masked) {  // uniform, rare
#pragma unroll
            for (int s = 0; s < 4; s++)
                if ((st_cm >> s) & 1u) {
                    if ((t.mask[plane * p + st_off[s] / 3u] >> (st_off[s] % 3u)) & 1u) v[s] = 0.0;
                }
        }
    };
    auto store_plane = [&](int buf, const double v[4]) {
#pragma unroll
        for (int s = 0; s < 4; s++) s_u[buf][tid + s * TILE * TILE] = v[s];
    };
    // the 4 in-plane nodes of this thread's element, natural order (lx + 2 ly), 2-D transformed
    const int o00 = (ty * TSTG + tx) * 3, o10 = o00 + 3, o01 = o00 + TSTG * 3, o11 = o01 + 3;
    double pre[4] = {0, 0, 0, 0};
    auto read_plane_wht = [&](int buf, double U[3][4]) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            U[c][0] = s_u[buf][o00 + c];
            U[c][1] = s_u[buf][o10 + c];
            U[c][2] = s_u[buf][o01 + c];
            U[c][3] = s_u[buf][o11 + c];
            wht4(U[c]);
        }
    };


    {   // all three planes in flight at once: one memory round trip instead of three
        double p0[4], p1[4];
        load_plane(kz0 - 1, p0);
        load_plane(kz0, p1);
        load_plane(kz0 + 1, pre);
        store_plane(0, p0);
        store_plane(1, p1);
        store_plane(2, pre);
    }
    __syncthreads();
    double Ub[3][4];
    read_plane_wht(0, Ub);
    double Cy[3][4];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int m = 0; m < 4; m++) Cy[c][m] = 0.0;
    double pdot = 0.0;
    double Elow = 0.0;  // DIAG_FLY: in-plane modulus sum of the previous element layer
    // MACRO: moduli of the 8 children of this thread's coarse element in layer l (0 outside the domain)
    auto load_children = [&](int l, double e8[8]) {
        const bool eok = elem_ok && l >= 0 && l < t.ezl;
#pragma unroll
        for (int ch = 0; ch < 8; ch += 2) {  // the x-pair of children is one aligned 16-byte load (fex is even)
            double2 v = make_double2(0.0, 0.0);
            if (eok) v = *(const double2 *)(t.E + ((long)(2 * ei) + (long)t.fex * ((2 * ej + ((ch >> 1) & 1)) + (long)t.fey * (2 * l + (ch >> 2)))));
            e8[ch] = v.x;
            e8[ch + 1] = v.y;
        }
    };
    double en[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (MACRO) load_children(kz0 - 1, en);

    for (int s = 0; s < nsteps; s++) {
        const int el = kz0 - 1 + s;  // element layer; bottom node plane el, top plane el+1
        const bool more = s + 1 < nsteps;
        const bool outp = s >= 1 && node_ok;
        // ---- issue the long-latency loads of this step first, in the order in which they are consumed (the
        // vector-memory counter retires in order): modulus (needed after the block products), epilogue operands
        // (after the barrier), next plane (end of the step)
        const int b0 = s % 3, b1 = (s + 1) % 3;  // ring slots of the bottom / top plane of this step
        double Eraw = 0.0;
        if (!MACRO && el >= 0 && el < t.ezl) Eraw = t.E[(long)t.ex * t.ey * el + eoff];  // uniform branch
        const long nq = 3 * (ncol + plane * el);
        double xo[3] = {0, 0, 0}, bo[3] = {0, 0, 0}, dd[3] = {0, 0, 0}, di[3] = {0, 0, 0}, co[3] = {0, 0, 0};
        if (outp) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                if (EPI == EPI_RESID || EPI == EPI_CHEB) bo[c] = a.b[nq + c];
                if (EPI == EPI_CHEB) {
                    if (!DIAG_FLY) {
                        dd[c] = a.d[nq + c];
                        di[c] = a.dinv[nq + c];
                    } else {
                        // 3-term form: the previous iterate lives in the output buffer (c1 = 0: first step, not read)
                        dd[c] = (a.c1 != 0.0 && !a.prev_zero) ? a.out[nq + c] : 0.0;
                    }
                }
                if (MACRO && t.corr) co[c] = t.corr[nq + c];
                if (EPI == EPI_APPLY && a.dinv) di[c] = a.dinv[nq + c];  // scaled product (the spectrum estimate's D^-1/2 A D^-1/2)
            }
        }
        if (more) load_plane(el + 3, pre);      // lands in slot b0 once this step is done with it
        if (outp) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                // own input value: the staged (masked) copy in LDS is exact unless the column carries a
                // Dirichlet condition (rare) -> no second trip to memory for x
                xo[c] = s_u[b0][o00 + c];
                if (tile_masked && own_cm) xo[c] = x[nq + c];
            }
        }
        // ---- element in the Walsh-Hadamard basis
        double Ut[3][4], u[3][8], f[3][8];
        read_plane_wht(b1, Ut);
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                u[c][m] = Ub[c][m] + Ut[c][m];
                u[c][m + 4] = Ub[c][m] - Ut[c][m];
                Ub[c][m] = Ut[c][m];
            }
        // opaque offset: keeps the 33 scalar loads inside the loop instead of 66 live SGPRs (hoisting them to the
        // top of the step measured no gain: the compiler spills the SGPRs to lanes)
        int boff;
        asm volatile("s_mov_b32 %0, %1" : "=s"(boff) : "s"(t.slot_off));
        double Ee = 0.0;
        if (!MACRO) {
            sym_ke_blocks(c_symB + boff, u, f);
            sym_ke_translation<KrylovEpi<EPI>::value>(c_symX + 4 * boff, u, f);
            Ee = Eraw * emul;
        } else {
            const bool eok = elem_ok && el >= 0 && el < t.ezl;
            Ee = eok ? 1.0 : 0.0;  // children moduli are applied inside
            double eh[8];
#pragma unroll
            for (int ch = 0; ch < 8; ch++) eh[ch] = en[ch];
            if (more) load_children(el + 1, en);  // one step ahead: the latency hides behind this step's arithmetic
            {   // Walsh-Hadamard transform of the 8 child moduli
                double lo[4] = {eh[0], eh[1], eh[2], eh[3]}, hi[4] = {eh[4], eh[5], eh[6], eh[7]};
                wht4(lo);
                wht4(hi);
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    eh[m] = lo[m] + hi[m];
                    eh[m + 4] = lo[m] - hi[m];
                }
            }
            // constants through the scalar cache (an LDS copy read with broadcast ds_read_b64 measured 15 % slower)
            int goff;
            asm volatile("s_mov_b32 %0, %1" : "=s"(goff) : "s"(t.macg_off));
            macg_apply(c_macG + goff, u, eh, f);
        }
        // ---- back to the two planes; the upper plane's part is carried (still transformed)
        double P[3][4];
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const double sum = f[c][m] + f[c][m + 4], dif = f[c][m] - f[c][m + 4];
                if (MACRO) {  // the children's moduli are inside f already (all zero for an element outside the domain)
                    P[c][m] = sum + Cy[c][m];
                    Cy[c][m] = dif;
                } else {
                    P[c][m] = fma(Ee, sum, Cy[c][m]);
                    Cy[c][m] = Ee * dif;
                }
            }
        double s0[3], s1[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            wht4(P[c]);  // nodal contributions of this element column to plane el
            s0[c] = P[c][0] + dpp_row_shr1(P[c][1]);  // node (ei, ej  ): own + left element
            s1[c] = P[c][2] + dpp_row_shr1(P[c][3]);  // node (ei, ej+1)
            s_y[s & 1][tid * 3 + c] = s1[c];
        }
        double ex2 = 0.0;
        if (DIAG_FLY) {
            ex2 = Ee + dpp_row_shr1(Ee);  // this element + its left neighbour
            s_e[s & 1][tid] = ex2;
        }
        __syncthreads();
        double e4 = 0.0;  // DIAG_FLY: the 4 elements of this layer around the node
        if (DIAG_FLY && node_ok) e4 = ex2 + s_e[s & 1][tid - TILE];
        if (outp) {
            unsigned m = 0;
            if (tile_masked && own_cm) m = t.mask[ncol + plane * el];
            if (DIAG_FLY) {
                const double rinv = 1.0 / (e4 + Elow);
#pragma unroll
                for (int c = 0; c < 3; c++) di[c] = ((m >> c) & 1u) ? 1.0 : rinv * c_symB[boff + SYMKE_N + c];
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                double y = s0[c] + s_y[s & 1][(tid - TILE) * 3 + c];
                if (MACRO) y += co[c];
                if ((m >> c) & 1u) y = xo[c];
                const long q = nq + c;
                if (EPI == EPI_APPLY) {
                    a.out[q] = a.dinv ? y * di[c] : y;
                } else if (EPI == EPI_RESID) {
                    a.out[q] = bo[c] - y;
                } else if (EPI == EPI_CHEB && DIAG_FLY) {
                    const double dprev = a.c1 != 0.0 ? xo[c] - dd[c] : 0.0;  // prev_zero: dd = 0
                    a.out[q] = xo[c] + (a.c1 * dprev + a.c2 * (di[c] * (bo[c] - y)));
                } else if (EPI == EPI_CHEB) {
                    const double dn = a.c1 * dd[c] + a.c2 * (di[c] * (bo[c] - y));
                    a.d[q] = dn;
                    a.out[q] = xo[c] + dn;
                } else {
                    a.out[q] = y;
                    pdot = fma(xo[c], y, pdot);
                }
            }
        }
        if (DIAG_FLY) Elow = e4;
        // slot b0 was last read before the barrier above; its next reader (step s+2) is behind the next one
        if (more) store_plane(b0, pre);
    }
    if (EPI == EPI_APPLY_DOT) {
        const double v[1] = {block_sum(pdot)};
        reduce_tail<1>(v, a.partials, gridDim.x * gridDim.y * gridDim.z, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z),
                       a.ticket, a.red_out);
    }
}

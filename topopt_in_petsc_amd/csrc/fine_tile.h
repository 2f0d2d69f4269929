// fine_tile.h -- second generation of the fine-level matrix-free hex8 operator (matfree_tile.h holds the
// mathematics: Walsh-Hadamard block form of KE, x/y/z combination of the element results without atomics).
//
// What changed against k_matfree_tile<.,0>, and why (ablation timings + counters of round 2):
//  1. EVERY global load has a whole step between issue and use, at no cost in registers, and the loop body is
//     STRAIGHT-LINE code.  The first kernel issued its loads at the top of a step and consumed them inside the same
//     step (57 % of the wave time parked).  Now: the node plane needed by step s+2 is requested at the top of
//     step s, sits in 4 VGPR pairs during the step and is written to a FOUR-slot LDS ring at the top of step s+1
//     (the extra slot is what lets the write move before the step's barrier); the modulus of layer s+1 is requested
//     at the top of step s; the epilogue operands of step s+1 are requested at the end of step s into the registers
//     the epilogue has just freed; the thread's own input value is read from the ring inside the epilogue.
//     The vector-memory counter retires in order, so the issue order is the consumption order:
//         [end of s-1] b, u-   [top of s] plane(s+3), E(s+1)   [epilogue s] out
//     Straight-line matters: the compiler's s_waitcnt insertion falls back to vmcnt(0) at every control-flow join
//     whose arms issued different numbers of memory operations.  So nothing in the loop is conditional: addresses
//     are clamped into the arrays (out-of-range planes / layers are zeroed by a select AFTER the load), the ring is
//     padded, the last two steps prefetch planes nobody reads, and the stores are predicated through a raw-buffer
//     descriptor (out-of-range offset = dropped by the hardware) instead of a branch.
//  2. (tried and dropped: re-dealing the epilogue vectors through LDS so that every memory instruction touches 16
//     consecutive doubles per tile row instead of 24-byte strided triples -- same time, more LDS traffic; the
//     line visits of the strided form are not what limits the kernel.)
//  3. Dirichlet tiles: the ring holds the TRUE values plus the mask bytes of the staged plane (in the slot's
//     padding); the mask is applied when the plane is read.  No global reads in the epilogue, two byte loads per
//     thread and step, only in the workgroups whose tile carries a condition.
#pragma once
#include <type_traits>

#include "matfree_tile.h"

#ifndef FT_ABL
#define FT_ABL 0  // ablation builds (tools/ablate_fine.sh): 1 no block products, 2 no plane loads, 3 no stores, 4 no barrier, 5 no E, 6 no epilogue loads, 8 one workgroup less per CU, 9 data movement only (no LDS, no barrier, no arithmetic)
#endif
#ifndef FT_STORE_AUX
#define FT_STORE_AUX 0  // cache policy bits of the output stores (gfx942+: 1 sc0, 2 sc1, 4 nt)
#endif
constexpr int RING = 4;
constexpr int SLOT = 4 * TILE * TILE;  // padded ring slot (STG_N = 867 used)

template <int EPI, bool MASKED>
__device__ __forceinline__ void fine_tile_run(const TileArgs &t, const NodeArgs &a, double (*s_u)[SLOT],
                                              double (*s_y)[TILE * TILE * 3], double (*s_e)[TILE * TILE], int bxi,
                                              int byi, int bzi) {
    // No implicit mul+add contraction in this function: the compiler peels / specialises iterations of the z-loop and
    // contracted them differently, which made a node's result depend (in the last bit) on whether its plane is the
    // first of a z-chunk.  With explicit fma() only, the result of a node is independent of the chunking -- which the
    // boundary-first launches of the halo overlap rely on (bitwise equal to the single launch).
#pragma clang fp contract(off)
    constexpr bool IS_CHEB = (EPI == EPI_CHEB || EPI == EPI_CHEB_DOT);
    constexpr bool DIAG_FLY = IS_CHEB;
    constexpr bool HAS_B = (EPI == EPI_RESID || IS_CHEB);
    const int tid = threadIdx.x;
    const int tx = tid & (TILE - 1), ty = tid / TILE;
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
    const long ncol = node_ok ? (long)ei + (long)t.nx * ej : 0;       // always a valid node column
    const int yprev = ty >= 1 ? tid - TILE : tid;                     // y-neighbour's slot (valid index for every lane)

    // staging slots of this thread: flat index f -> (row, node column, component); columns outside the domain are
    // clamped onto the boundary column (their values only reach elements outside the domain, whose modulus is 0)
    unsigned st_off[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int f = min(tid + s * TILE * TILE, STG_N - 1);
        const int r = f / (TSTG * 3), c = f % (TSTG * 3);
        const int gi = min(max(bx - 1 + c / 3, 0), t.nx - 1), gj = min(max(by - 1 + r, 0), t.ny - 1);
        st_off[s] = 3u * (unsigned)(gi + t.nx * gj) + (unsigned)(c % 3);
    }
    // EPI_APPLY_DOT: the product's input may be the CG direction in the making, staged = fma(beta, x, z) (NodeArgs::pz; one
    // rank).  ONE loop for both uses (two instantiations of this loop are scheduled differently by the compiler, see load_prev):
    // the plain product runs with z = x and beta = 0 -- fma(0, x, x) = x, the second load an L1 hit.
    constexpr bool FUSEP = (EPI == EPI_APPLY_DOT);
    const double *__restrict__ zin = (FUSEP && a.pz) ? a.pz : x;
    const double pbeta = (FUSEP && a.pz && a.pscal) ? a.pscal[a.slot_new] / a.pscal[a.slot_old] : 0.0;
    // plane p of the input (zeroed outside the slab when it is written to the ring); always 4 loads from valid addresses
    auto load_plane = [&](int p, double v[4]) {
        const double *__restrict__ xp = x + 3 * plane * min(max(p, 0), t.nzl - 1);
#pragma unroll
        for (int s = 0; s < 4; s++) v[s] = FT_ABL == 2 ? 1.0 + tid : xp[st_off[s]];
    };
    auto load_plane_z = [&](int p, double v[4]) {
        const double *__restrict__ zp = zin + 3 * plane * min(max(p, 0), t.nzl - 1);
#pragma unroll
        for (int s = 0; s < 4; s++) v[s] = zp[st_off[s]];
    };
    auto store_plane = [&](int buf, int p, const double v[4], const double vz[4]) {
        const bool inside = p >= 0 && p < t.nzl;  // uniform
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const double val = FUSEP ? fma(pbeta, v[s], vz[s]) : v[s];
            s_u[buf][min(tid + s * TILE * TILE, STG_N)] = inside ? val : 0.0;  // [STG_N]: dump slot
        }
    };
    // MASKED tiles: mask bytes of the 17 x 17 staged nodes of a plane, kept in the padding of the plane's ring slot
    unsigned mk_off[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int n = min(tid + s * TILE * TILE, TSTG * TSTG - 1);
        const int gi = min(max(bx - 1 + n % TSTG, 0), t.nx - 1), gj = min(max(by - 1 + n / TSTG, 0), t.ny - 1);
        mk_off[s] = (unsigned)(gi + t.nx * gj);
    }
    auto load_mask = [&](int p, unsigned mk[2]) {
        const uint8_t *__restrict__ mp = t.mask + plane * min(max(p, 0), t.nzl - 1);
        mk[0] = mp[mk_off[0]];
        mk[1] = mp[mk_off[1]];
    };
    auto slot_mask = [&](int buf) -> uint8_t * { return (uint8_t *)&s_u[buf][STG_N + 5]; };
    auto store_mask = [&](int buf, const unsigned mk[2]) {
        uint8_t *m = slot_mask(buf);
        m[tid] = (uint8_t)mk[0];
        m[min(tid + TILE * TILE, TSTG * TSTG - 1)] = (uint8_t)mk[1];  // lanes beyond the 289 nodes repeat node 288
    };
    const int o00 = (ty * TSTG + tx) * 3, o10 = o00 + 3, o01 = o00 + TSTG * 3, o11 = o01 + 3;
    const int n00 = ty * TSTG + tx;  // staged node index of the thread's own node (element corner 00)
    auto read_plane_wht = [&](int buf, double U[3][4]) {
        unsigned m4[4] = {0, 0, 0, 0};
        if (MASKED) {
            const uint8_t *m = slot_mask(buf);
            m4[0] = m[n00];
            m4[1] = m[n00 + 1];
            m4[2] = m[n00 + TSTG];
            m4[3] = m[n00 + TSTG + 1];
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            U[c][0] = s_u[buf][o00 + c];
            U[c][1] = s_u[buf][o10 + c];
            U[c][2] = s_u[buf][o01 + c];
            U[c][3] = s_u[buf][o11 + c];
            if (MASKED) {
#pragma unroll
                for (int q = 0; q < 4; q++) U[c][q] = ((m4[q] >> c) & 1u) ? 0.0 : U[c][q];
            }
            wht4(U[c]);
        }
    };
    const long lay = (long)t.ex * t.ey;
    auto load_E = [&](int l) -> double { return FT_ABL == 5 ? 1.0 : t.E[lay * min(max(l, 0), t.ezl - 1) + eoff]; };
    // epilogue operands of output plane `pl`: the 3 interlaced dofs of the thread's node (24 contiguous bytes)
    const long ncq = 3 * ncol;
    const unsigned voff_out = node_ok ? 8u * (unsigned)ncq : 0x7FFFFF00u;  // byte offset inside an output plane
    double bo[3] = {0, 0, 0}, dd[3] = {0, 0, 0};
    const bool read_prev = IS_CHEB && a.c1 != 0.0 && !a.prev_zero;  // uniform, loop invariant
    auto load_epi = [&](int pl) {
        const double *__restrict__ bp = a.b + 3 * plane * min(max(pl, 0), t.nzl - 1) + ncq;
#pragma unroll
        for (int c = 0; c < 3; c++) bo[c] = FT_ABL == 6 ? 1.0 : bp[c];
    };
    // 3-term Chebyshev: the previous iterate lives in the output buffer.  With c1 = 0 / a zero guess it is not needed --
    // but the loop is ONE piece of code for both cases (round 4): as two instantiations the compiler scheduled the one
    // without this load far worse (its four scalar loads of the block coefficients serialised through one SGPR range,
    // 21 waits per step: 72 us against 49-51 us back to back at 128^3, tools/cheb_variants.py).  A sweep's first step
    // therefore repeats the load of `b` here (same addresses as load_epi: served by the L1, no HBM bytes) and a
    // uniform select drops the value.
    const double *prev_base = read_prev ? a.out : a.b;
    auto load_prev = [&](int pl) {
        const double *pp = prev_base + 3 * plane * min(max(pl, 0), t.nzl - 1) + ncq;
#pragma unroll
        for (int c = 0; c < 3; c++) dd[c] = FT_ABL == 6 ? 1.0 : pp[c];
    };

    double pre[4], prez[4] = {0, 0, 0, 0};
    unsigned pmk[2] = {0, 0};
    double Enext;
    {   // prologue: planes j = 0, 1 -> ring, plane j = 2 -> registers (written at the top of step 0); one round trip
        double p0[4], p1[4], z0[4] = {0, 0, 0, 0}, z1[4] = {0, 0, 0, 0};
        unsigned m0[2] = {0, 0}, m1[2] = {0, 0};
        load_plane(kz0 - 1, p0);
        load_plane(kz0, p1);
        load_plane(kz0 + 1, pre);
        if (FUSEP) {
            load_plane_z(kz0 - 1, z0);
            load_plane_z(kz0, z1);
            load_plane_z(kz0 + 1, prez);
        }
        if (MASKED) {
            load_mask(kz0 - 1, m0);
            load_mask(kz0, m1);
            load_mask(kz0 + 1, pmk);
        }
        Enext = load_E(kz0 - 1);
        if (HAS_B) load_epi(kz0 - 1);
        if (IS_CHEB) load_prev(kz0 - 1);
        store_plane(0, kz0 - 1, p0, z0);
        store_plane(1, kz0, p1, z1);
        if (MASKED) {
            store_mask(0, m0);
            store_mask(1, m1);
        }
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
    double Elow = 0.0;

    auto step = [&](int s) {
        constexpr bool PREV = IS_CHEB;
        const int el = kz0 - 1 + s;  // element layer; bottom node plane el (ring slot s & 3), top el + 1
        if (FT_ABL == 9) {  // timing skeleton: same loads / stores, same pipeline distance, nothing else
            const double acc = (pre[0] + pre[1]) + (pre[2] + pre[3]) + Enext;
            load_plane(el + 3, pre);
            Enext = load_E(el + 1);
            double o9[3];
#pragma unroll
            for (int c = 0; c < 3; c++) o9[c] = acc + bo[c] + dd[c];
#pragma unroll
            for (int c = 0; c < 3; c++) asm volatile("" : "+v"(o9[c]));
            __builtin_amdgcn_sched_barrier(0);
            if (HAS_B) load_epi(el + 1);
            if (PREV) load_prev(el + 1);
            const __amdgpu_buffer_rsrc_t rs9 = __builtin_amdgcn_make_buffer_rsrc(a.out + 3 * plane * max(el, 0), 0, s >= 1 ? (int)(24 * plane) : 0, 0x00020000);
            typedef double d2_9 __attribute__((ext_vector_type(2)));
            typedef unsigned u4_9 __attribute__((ext_vector_type(4)));
            typedef unsigned u2_9 __attribute__((ext_vector_type(2)));
            const d2_9 q01 = {o9[0], o9[1]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_9, q01), rs9, voff_out, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_9, o9[2]), rs9, voff_out + 16u, 0, 0);
            return;
        }
        // ---- top of the step: retire last step's prefetch into the ring, request the next one
        store_plane((s + 2) & 3, el + 2, pre, prez);  // plane j = s + 2: read from step s + 1 on (behind this step's barrier)
        if (MASKED) store_mask((s + 2) & 3, pmk);
        const double Eraw = (el >= 0 && el < t.ezl) ? Enext : 0.0;
        load_plane(el + 3, pre);                     // plane j = s + 3: top plane of step s + 2
        if (FUSEP) load_plane_z(el + 3, prez);
        if (MASKED) load_mask(el + 3, pmk);
        Enext = load_E(el + 1);
        // ---- element in the Walsh-Hadamard basis
        double Ut[3][4], u[3][8], f[3][8];
        read_plane_wht((s + 1) & 3, Ut);
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                u[c][m] = Ub[c][m] + Ut[c][m];
                u[c][m + 4] = Ub[c][m] - Ut[c][m];
                Ub[c][m] = Ut[c][m];
            }
        int boff;
        asm volatile("s_mov_b32 %0, %1" : "=s"(boff) : "s"(t.slot_off));
        if (FT_ABL == 1) {
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int m = 0; m < 8; m++) f[c][m] = u[c][m];
        } else {
            sym_ke_blocks(c_symB + boff, u, f);
            sym_ke_translation<KrylovEpi<EPI>::value>(c_symX + 4 * boff, u, f);
        }
        const double Ee = Eraw * emul;
        double P[3][4];
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const double sum = f[c][m] + f[c][m + 4], dif = f[c][m] - f[c][m + 4];
                P[c][m] = fma(Ee, sum, Cy[c][m]);
                Cy[c][m] = Ee * dif;
            }
        double s0[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            wht4(P[c]);
            s0[c] = P[c][0] + dpp_row_shr1(P[c][1]);                           // node (ei, ej  ): own + left element
            s_y[s & 1][tid * 3 + c] = P[c][2] + dpp_row_shr1(P[c][3]);         // node (ei, ej+1)
        }
        double ex2 = 0.0;
        if (DIAG_FLY) {
            ex2 = Ee + dpp_row_shr1(Ee);
            s_e[s & 1][tid] = ex2;
        }
        if (FT_ABL != 4) __syncthreads();
        // ---- epilogue, computed by every lane; only the stores are predicated
        unsigned mown = 0;
        if (MASKED) mown = slot_mask(s & 3)[n00];
        double e4 = 0.0;
        if (DIAG_FLY) e4 = ex2 + s_e[s & 1][yprev];
        double di[3] = {0, 0, 0};
        if (DIAG_FLY) {
            const double rinv = 1.0 / (e4 + Elow);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                di[c] = rinv * c_symB[boff + SYMKE_N + c];
                if (MASKED) di[c] = ((mown >> c) & 1u) ? 1.0 : di[c];
            }
        }
        double o[3], xo3[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double xo = s_u[s & 3][o00 + c];  // own input value (the ring holds the true values)
            xo3[c] = xo;
            double y = s0[c] + s_y[s & 1][yprev * 3 + c];
            if (MASKED) y = ((mown >> c) & 1u) ? xo : y;
            if (EPI == EPI_APPLY) {
                o[c] = y;
            } else if (EPI == EPI_RESID) {
                o[c] = bo[c] - y;
            } else if (IS_CHEB) {
                const double dprev = read_prev ? xo - dd[c] : (a.c1 != 0.0 ? xo : 0.0);  // zero guess: u- = 0
                o[c] = xo + (a.c1 * dprev + a.c2 * (di[c] * (bo[c] - y)));
                if (EPI == EPI_CHEB_DOT) pdot = (s >= 1 && node_ok) ? fma(bo[c], o[c], pdot) : pdot;
            } else {
                o[c] = y;
                pdot = (s >= 1 && node_ok) ? fma(xo, y, pdot) : pdot;
            }
        }
        if (DIAG_FLY) Elow = e4;
        // operands of the next step's epilogue -- BEFORE this step's stores: the counter retires in order and nothing
        // ever waits for a store.  The pins and the scheduling barrier keep the compiler from hoisting these loads
        // above the uses of the registers they refill (it would load into fresh registers and copy at the loop end:
        // a wait on loads that were just issued).
#pragma unroll
        for (int c = 0; c < 3; c++) asm volatile("" : "+v"(o[c]));
        __builtin_amdgcn_sched_barrier(0);
        if (HAS_B) load_epi(el + 1);
        if (PREV) load_prev(el + 1);
        // Predication WITHOUT control flow (a branch around a store makes the compiler's count of outstanding memory
        // operations inexact, and every later wait degrades to vmcnt(0)): the plane is a raw buffer, lanes with
        // nothing to store carry an offset beyond its size and the hardware drops their stores; step 0 has size 0.
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            a.out + 3 * plane * max(el, 0), 0, (s >= 1 && FT_ABL != 3) ? (int)(24 * plane) : 0, 0x00020000);
        typedef double d2_t __attribute__((ext_vector_type(2)));
        typedef unsigned u4_t __attribute__((ext_vector_type(4)));
        typedef unsigned u2_t __attribute__((ext_vector_type(2)));
        const d2_t o01 = {o[0], o[1]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, o01), rs, voff_out, 0, FT_STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, o[2]), rs, voff_out + 16u, 0, FT_STORE_AUX);
        if (FUSEP) {  // the new CG direction of the thread's node (a descriptor of size 0 drops the stores of the plain product)
            double *pn = a.pnew ? a.pnew : a.out;
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(pn + 3 * plane * max(el, 0), 0, (s >= 1 && a.pnew) ? (int)(24 * plane) : 0, 0x00020000);
            const d2_t p01 = {xo3[0], xo3[1]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, p01), rp, voff_out, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, xo3[2]), rp, voff_out + 16u, 0, 0);
        }
    };
    for (int s = 0; s < nsteps; s++) step(s);
    if (EPI == EPI_APPLY_DOT || EPI == EPI_CHEB_DOT) {
        const double v[1] = {block_sum(pdot)};
        reduce_tail<1>(v, a.partials, gridDim.x * gridDim.y * gridDim.z, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z),
                       a.ticket, a.red_out);
    }
}

// The Krylov product carries 135 more scalar-loaded constants through its loop: at three workgroups per CU (168 VGPRs) the
// compiler spills 52 vector and 165 scalar registers into it (79.9 us at 128^3), at two none (59.5 us; the packed form: 36.6).
#ifdef SYMKE_X_OCC3
#define FT_WAVES(EPI) 3
#else
#define FT_WAVES(EPI) (KrylovEpi<EPI>::value ? 2 : 3)
#endif
template <int EPI>
__global__ __launch_bounds__(TILE * TILE, FT_WAVES(EPI)) void k_fine_tile(TileArgs t, NodeArgs a) {
    __shared__ double s_u[RING][SLOT];           // node-plane ring
    __shared__ double s_y[2][TILE * TILE * 3];   // y-combination, double buffered -> one barrier per step
    __shared__ double s_e[2][TILE * TILE];       // modulus sums for the on-the-fly Jacobi diagonal (CHEB)
#if FT_ABL == 8
    __shared__ double s_pad[4096];  // 32 KB more: two workgroups per CU
    if (t.nx < 0) s_pad[threadIdx.x] = t.kz, a.out[0] = s_pad[255 - threadIdx.x];
#endif
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (private L2 each); give every XCD a
    // contiguous run of tiles so that neighbouring tiles, which share halo columns, meet in one L2
    int bxi, byi, bzi;
    {
        const int nb = gridDim.x * gridDim.y * gridDim.z;
        const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int x8 = lin & 7;
        const int m = t.xcd_remap ? x8 * (nb >> 3) + min(x8, nb & 7) + (lin >> 3) : lin;  // bijection of [0, nb)
        bxi = m % gridDim.x;
        byi = (m / gridDim.x) % gridDim.y;
        bzi = m / (gridDim.x * gridDim.y);
    }
    // does this tile (staged columns) carry a Dirichlet condition anywhere?  workgroup-uniform
    bool masked = false;
    if (t.colmask) {
        unsigned any = 0;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int f = min((int)threadIdx.x + s * TILE * TILE, STG_N - 1);
            const int r = f / (TSTG * 3), c = f % (TSTG * 3);
            const int gi = min(max(bxi * TOUT - 1 + c / 3, 0), t.nx - 1), gj = min(max(byi * TOUT - 1 + r, 0), t.ny - 1);
            any |= t.colmask[gi + t.nx * gj];
        }
        masked = __syncthreads_or(any != 0u) != 0;
    }
    if (masked)
        fine_tile_run<EPI, true>(t, a, s_u, s_y, s_e, bxi, byi, bzi);
    else
        fine_tile_run<EPI, false>(t, a, s_u, s_y, s_e, bxi, byi, bzi);
}

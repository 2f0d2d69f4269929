// fine_u4.h -- the unmasked tiles of the fine-level operator, written for the INSTRUCTION COUNT of a step.
//
// What the probe runs of round 3 showed (tools/probe/fine_probe.hip): k_fine_tile and k_fine_dma spend the
// same time per step although one waits for memory and the other does not -- 2.2-2.7 us per step with three waves per
// SIMD, which is 3 waves x ~450 instructions x 4 cycles: a SIMD of this chip starts about ONE instruction per 4-cycle
// turn, whatever its type.  The "memory-free" ablation of k_fine_dma (every descriptor empty) takes 234 us at 256^3,
// the data-movement skeleton 245 us, the real kernel 295 us.  So the step is rewritten around its instruction count
// (447 -> ~280 for the plain product):
//   * the step loop is unrolled by the ring length (4 planes / 4 modulus layers): slots, the parity of the 16-byte
//     staging windows and the LDS-DMA destinations are compile-time properties of a position, every LDS address of a
//     lane is one of two precomputed registers plus an immediate;
//   * descriptors advance by an addition per step; the "is this plane needed / inside the array" logic of k_fine_dma is
//     one subtraction from a limit address and a max with 0;
//   * the transformed top plane is carried to the next step (CARRY) where the registers allow it.
// Queue discipline, hazards and waits: fine_dma.h (D = 2).
// Tiles that carry a Dirichlet condition (MASKED: one tile column in nine of a cantilever) take the mask bytes of their
// four in-plane nodes straight into registers, one step ahead, with hand-counted loads like the epilogue operands:
// + 4 loads and ~60 integer instructions per step.  (Staging them through LDS as fine_dma.h does costs 5 KB of LDS,
// which is the third workgroup per CU; the compiler-counted loads of fine_dma.h drained the DMA queue every step:
// 243 -> 307 us for the 256^3 product with only the x = 0 face clamped.)
#pragma once
#include "fine_dma.h"

template <int TX, int TY>
struct FineU4 {
    using B = FineDma<TX, TY, 2>;
    static constexpr int NT = B::NT, NW = B::NW, TOX = B::TOX, TOY = B::TOY;
    static constexpr int RING = 4;
    static constexpr int OFF_U = 0;
    static constexpr int OFF_E = OFF_U + RING * B::USLOT;
    static constexpr int OFF_Y = OFF_E + RING * B::ESLOT;
    static constexpr int OFF_RED = OFF_Y;
    static constexpr int LDS_OWN = OFF_Y + 2 * B::YBUF;
    static constexpr int LDS_BYTES = LDS_OWN;
};

#ifndef FU_COL0_FIRST
#define FU_COL0_FIRST 0  // measured at 256^3 with the x = 0 face clamped: 32 x 8 tiles 260 us with, 245 us without; 16 x 16: 253 / 274 -- inside the run-to-run spread, off
#endif
typedef unsigned fu_u4 __attribute__((ext_vector_type(4)));

// LDS-DMA with the descriptor as four plain words; M0 = a + b inside the statement (no save / restore: nothing else in
// these kernels uses M0); s_nop 3 + the addition = the 5 wait states between an SGPR write and its vector-memory reader
__device__ __forceinline__ void fu_dma16(unsigned m0a, unsigned m0b, unsigned voff, fu_u4 rs) {
    asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 3\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds" ::"s"(m0a), "s"(m0b), "v"(voff), "s"(rs) : "memory");
}

template <int EPI, int TX, int TY, bool CARRY, bool MASKED, bool BITOPS = true>
// experiment builds (make EXTRA='-DFU4_ST_MOD="\" nt\""' OUT=../libtopopt_amd_x.so, loaded through TP_LIB): cache-policy modifier
// of the output stores / of the once-streamed epilogue operands (b, previous iterate).  Measured at 256^3 in round 5: nt stores
// 251.2 / 391.3 us (product / Chebyshev step) against 249.3 / 392.6 -- nothing; nt on the epilogue loads as well 254.2 / 441.7 --
// the previous iterate is read back through the L2 by the same workgroup, the hint throws that away.  Default: none.
#ifndef FU4_ST_MOD
#define FU4_ST_MOD ""
#endif
#ifndef FU4_LD_MOD
#define FU4_LD_MOD ""
#endif
__device__ __forceinline__ void fine_u4_run(const TileArgs &t, const NodeArgs &a, char *lds, int bxi, int byi, int bzi) {
#pragma clang fp contract(off)
    using S = FineU4<TX, TY>;
    using B = typename S::B;
    constexpr bool IS_CHEB = (EPI == EPI_CHEB || EPI == EPI_CHEB_DOT);
    constexpr bool DIAG_FLY = IS_CHEB;
    constexpr bool HAS_B = (EPI == EPI_RESID || IS_CHEB);
    constexpr bool IS_DOT = (EPI == EPI_APPLY_DOT || EPI == EPI_CHEB_DOT);
    constexpr bool NEED_XO = IS_CHEB || EPI == EPI_APPLY_DOT || MASKED;
    constexpr int NMK = MASKED ? 4 : 0;  // mask loads per lane and step
    constexpr int NIU_W = B::NIU_W, NIE_W = B::NIE_W, NI = NIU_W + NIE_W;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tx = tid % TX, ty = tid / TX;
    const int bx = bxi * S::TOX, by = byi * S::TOY;
    int kz0, kz1;
    tile_chunk(t, bzi, kz0, kz1);
    const int nsteps = kz1 - kz0 + 2;
    const int ei = bx - 1 + tx, ej = by - 1 + ty;
    const bool elem_ok = ei >= 0 && ei < t.ex && ej >= 0 && ej < t.ey;
    const bool node_ok = tx >= 1 && ty >= 1 && ei < t.nx && ej < t.ny;
    const long plane = (long)t.nx * t.ny;
    const long lay = (long)t.ex * t.ey;
    const long ncol = node_ok ? (long)ei + (long)t.nx * ej : 0;
    const int yprev = ty >= 1 ? tid - TX : tid;
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char *)lds;
    const bool live = __builtin_amdgcn_readfirstlane((int)(__ballot(elem_ok || node_ok || ej < 0) != 0ull)) != 0;

    // ---- parities of the staging windows: plane jj of the chunk starts (rem) doubles above a 16-byte boundary,
    // rem(jj) = Ru[jj & 1]; the same for the modulus layers
    const int su_odd = (int)((3 * plane) & 1), se_odd = (int)(lay & 1);
    int Ru[2], Re[2];
    {
        const long cu = (long)((unsigned long)a.x >> 3) + 3 * plane * (long)(kz0 - 1), ce = (long)((unsigned long)t.E >> 3) + lay * (long)(kz0 - 1);
        Ru[0] = (int)(cu & 1), Ru[1] = Ru[0] ^ su_odd;
        Re[0] = (int)(ce & 1), Re[1] = Re[0] ^ se_odd;
    }
    // ---- DMA units of this lane (fine_dma.h), wave-major: wave w issues instructions w * NIU_W .. of a plane
    unsigned VU[2][NIU_W], VE[2][NIE_W];
    unsigned m0U[NIU_W], m0E[NIE_W];
#pragma unroll
    for (int k = 0; k < NIU_W; k++) {
        const int j = min(wave * NIU_W + k, B::NIU - 1);
        const int w = j * 64 + lane;
        m0U[k] = lds0 + S::OFF_U + (unsigned)(j * 1024);
        const int r = w / B::UPR_U, q = w % B::UPR_U;
        const int gj = min(max(by - 1 + r, 0), t.ny - 1);
        const int c = 3 * (bx - 1 + t.nx * gj);
        const unsigned st = w < B::NUNIT_U ? ((unsigned)(8 * (c & ~1) + 16 * q) | (unsigned)(c & 1)) : FD_OOB;
        VU[0][k] = fd_voff(st, Ru[0]), VU[1][k] = fd_voff(st, Ru[1]);
    }
#pragma unroll
    for (int k = 0; k < NIE_W; k++) {
        const int j = min(wave * NIE_W + k, B::NIE - 1);
        const int w = j * 64 + lane;
        m0E[k] = lds0 + S::OFF_E + (unsigned)(j * 1024);
        const int r = w / B::UPR_E, q = w % B::UPR_E;
        const int gj = min(max(by - 1 + r, 0), t.ey - 1);
        const int c = bx - 1 + t.ex * gj;
        const unsigned st = w < B::NUNIT_E ? ((unsigned)(8 * (c & ~1) + 16 * q) | (unsigned)(c & 1)) : FD_OOB;
        VE[0][k] = fd_voff(st, Re[0]), VE[1][k] = fd_voff(st, Re[1]);
    }
    // ---- LDS byte offsets of this lane inside a slot, per parity: own row / next row of the node planes, own modulus,
    // (Chebyshev) the modulus row below
    const int oddU0 = (bx - 1 + t.nx * ej) & 1, oddU1 = oddU0 ^ (t.nx & 1);
    const int oddE = (bx - 1 + t.ex * ej) & 1, oddEd = oddE ^ (t.ex & 1);
    unsigned AU0[2], AU1[2], AE[2], AEd[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        AU0[p] = (unsigned)((ty * B::ROWW_U + 3 * tx) * 8 + 8 * (Ru[p] ^ oddU0));
        AU1[p] = (unsigned)(((ty + 1) * B::ROWW_U + 3 * tx) * 8 + 8 * (Ru[p] ^ oddU1));
        AE[p] = (unsigned)((ty * B::ROWW_E + tx) * 8 + 8 * (Re[p] ^ oddE));
        AEd[p] = (unsigned)(((ty - 1) * B::ROWW_E + tx) * 8 + 8 * (Re[p] ^ oddEd)) + (unsigned)B::ESLOT * 4u;  // kept non-negative: the slot term is subtracted again
    }
    const bool okL = tx >= 1 && ei - 1 >= 0 && ei - 1 < t.ex && ej >= 0 && ej < t.ey;
    const bool okD = ty >= 1 && ei >= 0 && ei < t.ex && ej - 1 >= 0 && ej - 1 < t.ey;
    const bool okDL = tx >= 1 && ty >= 1 && ei - 1 >= 0 && ei - 1 < t.ex && ej - 1 >= 0 && ej - 1 < t.ey;

    // ---- running pointers (bytes) and limits.  A descriptor is {base, 0, max(limit - base, 0), flags}: everything a
    // chunk does not need, and everything beyond the arrays, lies at or above the limit.
    const unsigned long su = 24ul * (unsigned long)plane, se = 8ul * (unsigned long)lay;
    const unsigned long x0 = (unsigned long)a.x, E0 = (unsigned long)t.E;
    unsigned long pu = x0 + su * (unsigned long)(kz0 + 2);   // plane jj = 3: the first plane the loop requests
    unsigned long pe = E0 + se * (unsigned long)(kz0 + 1);   // layer jl = 2
    const unsigned long lim_u = x0 + su * (unsigned long)(min(kz1 + 1, t.nzl - 1) + 1);
    const unsigned long lim_e = E0 + se * (unsigned long)(min(kz1, t.ezl - 1) + 1);
    unsigned long pb = (unsigned long)a.b + su * (unsigned long)kz0, pp = (unsigned long)a.out + su * (unsigned long)kz0;
    unsigned long po = (unsigned long)a.out + su * (unsigned long)kz0 - su;  // plane kz0 - 1 (never stored)
    const unsigned long lim_o = (unsigned long)a.out + su * (unsigned long)(kz1 + 1), lim_b = (unsigned long)a.b + su * (unsigned long)(kz1 + 1);
    const bool read_prev = IS_CHEB && a.c1 != 0.0 && !a.prev_zero;
    const unsigned long lim_p = read_prev ? lim_o : 0ul;
    auto desc = [&](unsigned long base, unsigned long lim) -> fu_u4 {
        const int room = (int)((unsigned)lim - (unsigned)base);
        fu_u4 d = {(unsigned)base, (unsigned)(base >> 32), (unsigned)max(room, 0), FD_RSRC_FLAGS};
        return d;
    };
    auto desc_lim0 = [&](unsigned long base, unsigned long lim) -> fu_u4 {  // lim == 0: always empty
        const int room = lim ? (int)((unsigned)lim - (unsigned)base) : 0;
        fu_u4 d = {(unsigned)base, (unsigned)(base >> 32), (unsigned)max(room, 0), FD_RSRC_FLAGS};
        return d;
    };

    // ---- prologue: planes jj = 0, 1, 2 and layers -1 (empty), 0, 1 in the general form, with the queue pattern of a step
    const unsigned voff_out = node_ok ? 24u * (unsigned)ncol : 0x7FFFFF00u;
    fd_d2 b01 = {0.0, 0.0}, p01 = {0.0, 0.0};
    double b2 = 0.0, p2 = 0.0;
    constexpr int NLE = HAS_B ? (IS_CHEB ? 4 : 2) : 0;
    constexpr int OPS = NI + NMK + NLE + 2;
    constexpr int W_MID = OPS + NMK, W_PRO = MASKED ? 0 : (NLE + 2) + OPS, W_EPI = 2 + NI + NMK;
    const fu_u4 empty = {(unsigned)x0, (unsigned)(x0 >> 32), 0u, FD_RSRC_FLAGS};
    auto stores = [&](fu_u4 d, const double o[3]) {
        const fd_d2 o01 = {o[0], o[1]};
        asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %2, %3, 0 offen" FU4_ST_MOD "\n\tbuffer_store_dwordx2 %1, %2, %3, 0 offen offset:16" FU4_ST_MOD "\n\ts_nop 1" ::"v"(o01), "v"(o[2]),
                     "v"(voff_out), "s"(d)
                     : "memory");
    };
    const double zero3[3] = {0.0, 0.0, 0.0};
    const double *xend = a.x + 3 * plane * t.nzl, *Eend = t.E + lay * t.ezl;
    // MASKED: byte offsets of the element's four nodes inside a mask plane (clamped into the plane: the mask of a node
    // outside the domain meets a zero modulus), the running plane pointer, the four bytes in flight
    unsigned VM[4] = {0, 0, 0, 0}, mk[4] = {0, 0, 0, 0};
    unsigned long pm = 0, lim_m = 0;
    if (MASKED) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int gi = min(max(ei + (q & 1), 0), t.nx - 1), gj = min(max(ej + (q >> 1), 0), t.ny - 1);
            VM[q] = (unsigned)(gi + t.nx * gj);
        }
        pm = (unsigned long)t.mask + (unsigned long)plane * (unsigned long)(kz0 + 1);  // plane jj = 2: the first one the loop requests
        lim_m = (unsigned long)t.mask + (unsigned long)plane * (unsigned long)t.nzl;
    }
    auto mask_loads = [&](fu_u4 d) {
        asm volatile("s_nop 4\n\tbuffer_load_ubyte %0, %4, %8, 0 offen\n\tbuffer_load_ubyte %1, %5, %8, 0 offen\n\tbuffer_load_ubyte %2, %6, %8, 0 offen\n\tbuffer_load_ubyte %3, %7, %8, 0 offen"
                     : "+v"(mk[0]), "+v"(mk[1]), "+v"(mk[2]), "+v"(mk[3])
                     : "v"(VM[0]), "v"(VM[1]), "v"(VM[2]), "v"(VM[3]), "s"(d)
                     : "memory");
    };
    auto mask_pack = [&]() -> unsigned { return (mk[0] & 7u) | ((mk[1] & 7u) << 3) | ((mk[2] & 7u) << 6) | ((mk[3] & 7u) << 9); };
    unsigned mbot = 0, mtop = 0;
    for (int tb = -3; tb <= -1; tb++) {
        const int jj = tb + 3, jl = tb + 2;
        {
            const int p = kz0 - 1 + jj, pc = min(max(p, 0), t.nzl - 1);
            int rem;
            const __amdgpu_buffer_rsrc_t rs = fd_window(a.x + 3 * plane * pc, xend, p == pc && jj <= nsteps, &rem);
#pragma unroll
            for (int k = 0; k < NIU_W; k++) fd_dma16(m0U[k] + (unsigned)((jj & 3) * B::USLOT), (jj & 1) ? VU[1][k] : VU[0][k], rs);
        }
        {
            const int l = kz0 - 1 + jl, lc = min(max(l, 0), t.ezl - 1);
            int rem;
            const __amdgpu_buffer_rsrc_t rs = fd_window(t.E + lay * lc, Eend, l == lc && jl >= 0 && jl < nsteps, &rem);
#pragma unroll
            for (int k = 0; k < NIE_W; k++) fd_dma16(m0E[k] + (unsigned)((jl & 3) * B::ESLOT), (jl & 1) ? VE[1][k] : VE[0][k], rs);
        }
#pragma unroll
        for (int q = 0; q < NLE / 2 + 1; q++) stores(empty, zero3);
    }
    if (MASKED) {
        const int p0 = min(max(kz0 - 1, 0), t.nzl - 1), p1 = min(max(kz0, 0), t.nzl - 1);
        mask_loads(desc((unsigned long)t.mask + (unsigned long)plane * (unsigned long)p0, lim_m));
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(mk[0]), "+v"(mk[1]), "+v"(mk[2]), "+v"(mk[3])::"memory");
        mbot = mask_pack();
        mask_loads(desc((unsigned long)t.mask + (unsigned long)plane * (unsigned long)p1, lim_m));
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(mk[0]), "+v"(mk[1]), "+v"(mk[2]), "+v"(mk[3])::"memory");
    }
    fd_wait<W_PRO>();
    __syncthreads();

    double Ub[3][4], Cy[3][4];
    auto read_plane = [&](const char *sp, unsigned a0, unsigned a1, unsigned m12, double U[3][4], double x0v[3]) {
        const double *r0 = (const double *)(sp + a0), *r1 = (const double *)(sp + a1);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            U[c][0] = r0[c], U[c][1] = r0[3 + c], U[c][2] = r1[c], U[c][3] = r1[3 + c];
            x0v[c] = U[c][0];
            if (MASKED) {
                // clamped dof -> +0.0.  BITOPS: (sign-extended "not clamped" bit) & both words -- no compare, no SGPR pair
                // per value (the select form spills hundreds of SGPRs) but more vector registers: the form of the kernels
                // built for two waves per SIMD; those built for three keep the select (the bit form spills VGPRs there:
                // 260 -> 340 us for the 256^3 product).
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (BITOPS) {
                        const int keep = ((int)(~m12 << (31 - (3 * q + c)))) >> 31;
                        U[c][q] = __hiloint2double(__double2hiint(U[c][q]) & keep, __double2loint(U[c][q]) & keep);
                    } else {
                        U[c][q] = ((m12 >> (3 * q + c)) & 1u) ? 0.0 : U[c][q];
                    }
                }
            }
            wht4(U[c]);
        }
    };
    if (CARRY && live) {
        double xv[3];
        read_plane(lds + S::OFF_U, AU0[0], AU1[0], mbot, Ub, xv);
    }
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int m = 0; m < 4; m++) Cy[c][m] = 0.0;
    double pdot = 0.0, Elow = 0.0;
    double(*s_y)[(S::NT - TX) * 3] = (double(*)[(S::NT - TX) * 3])(lds + S::OFF_Y);
    int boff;
    asm volatile("s_mov_b32 %0, %1" : "=s"(boff) : "s"(t.slot_off));

    // one step at ring position Q (s = Q mod 4); IDLE: a wave without a part of the domain (queue and barriers only)
    auto step = [&](auto qc, auto prevc, auto idlec, int s) {
        constexpr int Q = decltype(qc)::value;
        constexpr bool PREV = decltype(prevc)::value, IDLE = decltype(idlec)::value;
        constexpr int PB = Q & 1, PT = (Q + 1) & 1;  // parity of the bottom plane / layer (jj = s) and of the top plane
        if (MASKED) {  // the bytes of plane s + 1 (requested at the top of the previous step) -> mtop; nothing younger than its tail
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(mk[0]), "+v"(mk[1]), "+v"(mk[2]), "+v"(mk[3]) : "n"(NLE + 2) : "memory");
            mtop = mask_pack();
        }
        // ---- batch(s): plane jj = s + 3 (parity PT, slot Q + 3), layer jl = s + 2 (parity PB, slot Q + 2)
        {
            const fu_u4 du = desc(pu & ~15ul, lim_u);
#pragma unroll
            for (int k = 0; k < NIU_W; k++) fu_dma16(m0U[k], (unsigned)(((Q + 3) & 3) * B::USLOT), VU[PT][k], du);
            pu += su;
            const fu_u4 de = desc(pe & ~15ul, lim_e);
#pragma unroll
            for (int k = 0; k < NIE_W; k++) fu_dma16(m0E[k], (unsigned)(((Q + 2) & 3) * B::ESLOT), VE[PB][k], de);
            pe += se;
            if (MASKED) {
                mask_loads(desc(pm, lim_m));  // plane jj = s + 2
                pm += (unsigned long)plane;
            }
        }
        double xo[3] = {0.0, 0.0, 0.0}, s0[3] = {0.0, 0.0, 0.0}, e4 = 0.0;
        if (!IDLE) {
            const char *sb = lds + S::OFF_U + Q * B::USLOT, *st = lds + S::OFF_U + ((Q + 1) & 3) * B::USLOT, *sE = lds + S::OFF_E + Q * B::ESLOT;
            const double Eraw = *(const double *)(sE + AE[PB]);
            double Ut[3][4], xt[3], u[3][8], f[3][8];
            if (!CARRY) {
                read_plane(sb, AU0[PB], AU1[PB], mbot, Ub, xo);
            } else if (NEED_XO) {
                const double *r0 = (const double *)(sb + AU0[PB]);
#pragma unroll
                for (int c = 0; c < 3; c++) xo[c] = r0[c];
            }
            read_plane(st, AU0[PT], AU1[PT], mtop, Ut, xt);
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    u[c][m] = Ub[c][m] + Ut[c][m];
                    u[c][m + 4] = Ub[c][m] - Ut[c][m];
                    if (CARRY) Ub[c][m] = Ut[c][m];
                }
            sym_ke_blocks(c_symB + boff, u, f);
            sym_ke_translation<KrylovEpi<EPI>::value>(c_symX + 4 * boff, u, f);
            const double Ee = elem_ok ? Eraw : 0.0;
            double P[3][4];
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const double sum = f[c][m] + f[c][m + 4], dif = f[c][m] - f[c][m + 4];
                    P[c][m] = fma(Ee, sum, Cy[c][m]);
                    Cy[c][m] = Ee * dif;
                }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                wht4(P[c]);
                s0[c] = P[c][0] + dpp_left<TX>(P[c][1]);
                const double up = P[c][2] + dpp_left<TX>(P[c][3]);
                if (ty < TY - 1) s_y[Q & 1][tid * 3 + c] = up;
            }
            if (DIAG_FLY) {
                const double *own = (const double *)(sE + AE[PB]), *dn = (const double *)(sE + AEd[PB] - B::ESLOT * 4);
                const double eL = own[-1], eD = dn[0], eDL = dn[-1];
                const double ex2 = Ee + (okL ? eL : 0.0);
                e4 = ex2 + ((okD ? eD : 0.0) + (okDL ? eDL : 0.0));
            }
        }
        fd_wait<W_MID>();
        __syncthreads();
        if (HAS_B) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(b01), "+v"(b2), "+v"(p01), "+v"(p2) : "n"(W_EPI));
        double o[3] = {0.0, 0.0, 0.0};
        if (!IDLE) {
            double di[3] = {0, 0, 0};
            if (DIAG_FLY) {
                const double rinv = 1.0 / (e4 + Elow);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    di[c] = rinv * c_symB[boff + SYMKE_N + c];
                    if (MASKED) di[c] = ((mbot >> c) & 1u) ? 1.0 : di[c];  // (own node = corner 0 of the bottom plane)
                }
                Elow = e4;
            }
            const double bo[3] = {b01.x, b01.y, b2}, dd[3] = {p01.x, p01.y, p2};
#pragma unroll
            for (int c = 0; c < 3; c++) {
                double y = s0[c] + s_y[Q & 1][yprev * 3 + c];
                if (MASKED) y = ((mbot >> c) & 1u) ? xo[c] : y;
                if (EPI == EPI_APPLY) {
                    o[c] = y;
                } else if (EPI == EPI_RESID) {
                    o[c] = bo[c] - y;
                } else if (IS_CHEB) {
                    const double dprev = PREV ? xo[c] - dd[c] : (a.c1 != 0.0 ? xo[c] : 0.0);
                    o[c] = xo[c] + (a.c1 * dprev + a.c2 * (di[c] * (bo[c] - y)));
                    if (EPI == EPI_CHEB_DOT) pdot = (s >= 1 && node_ok) ? fma(bo[c], o[c], pdot) : pdot;
                } else {
                    o[c] = y;
                    pdot = (s >= 1 && node_ok) ? fma(xo[c], y, pdot) : pdot;
                }
            }
#pragma unroll
            for (int c = 0; c < 3; c++) asm volatile("" : "+v"(o[c]));
        }
        // ---- operands of step s + 1 (planes kz0 + s), then the stores of step s (plane kz0 - 1 + s; s = 0: nothing)
        if (HAS_B) {
            const fu_u4 db = desc(pb, lim_b);
            asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, 0 offen" FU4_LD_MOD "\n\tbuffer_load_dwordx2 %1, %2, %3, 0 offen offset:16" FU4_LD_MOD
                         : "+v"(b01), "+v"(b2)
                         : "v"(voff_out), "s"(db)
                         : "memory");
            pb += su;
            if (IS_CHEB) {
                const fu_u4 dp = desc_lim0(pp, lim_p);
                asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, 0 offen" FU4_LD_MOD "\n\tbuffer_load_dwordx2 %1, %2, %3, 0 offen offset:16" FU4_LD_MOD
                             : "+v"(p01), "+v"(p2)
                             : "v"(voff_out), "s"(dp)
                             : "memory");
                pp += su;
            }
        }
        {
            fu_u4 dst = desc(po, lim_o);
            if (Q == 0) dst.z = s >= 1 ? dst.z : 0u;  // (s = 0 only happens at position 0)
            stores(dst, o);
            po += su;
        }
        if (MASKED) mbot = mtop;
    };
    auto run = [&](auto prevc, auto idlec) {
        int s = 0;
        while (true) {
            step(std::integral_constant<int, 0>{}, prevc, idlec, s);
            if (++s >= nsteps) break;
            step(std::integral_constant<int, 1>{}, prevc, idlec, s);
            if (++s >= nsteps) break;
            step(std::integral_constant<int, 2>{}, prevc, idlec, s);
            if (++s >= nsteps) break;
            step(std::integral_constant<int, 3>{}, prevc, idlec, s);
            if (++s >= nsteps) break;
        }
    };
    if (!live)
        run(std::false_type{}, std::true_type{});
    else if (read_prev)
        run(std::true_type{}, std::false_type{});
    else
        run(std::false_type{}, std::false_type{});
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(b01), "+v"(b2), "+v"(p01), "+v"(p2), "+v"(mk[0]), "+v"(mk[1]), "+v"(mk[2]), "+v"(mk[3])::"memory");
    if (IS_DOT) {
        double v = wave_sum(pdot);
        double *s_red = (double *)(lds + S::OFF_RED);
        __syncthreads();
        if (lane == 0) s_red[wave] = v;
        __syncthreads();
        double tot = 0.0;
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < S::NW; i++) tot += s_red[i];
        }
        const double vv[1] = {tot};
        reduce_tail<1>(vv, a.partials, gridDim.x * gridDim.y * gridDim.z, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z),
                       a.ticket, a.red_out);
    }
}

template <int EPI, int TX, int TY, int WPS, bool CARRY>
__global__ __launch_bounds__(TX *TY, WPS) void k_fine_u4(TileArgs t, NodeArgs a) {
    using S = FineU4<TX, TY>;
    __shared__ __attribute__((aligned(1024))) char lds[S::LDS_BYTES];
    int bxi, byi, bzi;
    {
        const int nb = gridDim.x * gridDim.y * gridDim.z;
        const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int x8 = lin & 7;
        int m = t.xcd_remap ? x8 * (nb >> 3) + min(x8, nb & 7) + (lin >> 3) : lin;
        // The tile column at x = 0 first: the face every load case of the reference clamps (LinearElasticity.cc:153-157)
        // lies there, its tiles take ~20 % longer (masks) and a launch of several rounds ends with whatever started last.
        const int gx = gridDim.x, gy = gridDim.y, n0 = gridDim.y * gridDim.z;
        if (FU_COL0_FIRST && gx > 1) {
            if (m < n0) {
                bxi = 0, byi = m % gy, bzi = m / gy;
            } else {
                m -= n0;
                bxi = 1 + m % (gx - 1), byi = (m / (gx - 1)) % gy, bzi = m / ((gx - 1) * gy);
            }
        } else {
            bxi = m % gx, byi = (m / gx) % gy, bzi = m / (gx * gy);
        }
    }
    bool masked = false;
    if (t.colmask) {
        unsigned any = 0;
        for (int n = threadIdx.x; n < S::B::UROWS * S::B::UPTS; n += S::NT) {
            const int gi = min(max(bxi * S::TOX - 1 + n % S::B::UPTS, 0), t.nx - 1), gj = min(max(byi * S::TOY - 1 + n / S::B::UPTS, 0), t.ny - 1);
            any |= t.colmask[gi + t.nx * gj];
        }
        masked = __builtin_amdgcn_readfirstlane(__syncthreads_or(any != 0u)) != 0;
    }
    if (masked)
        fine_u4_run<EPI, TX, TY, (WPS * TX * TY <= 512), true, (WPS * TX * TY <= 512)>(t, a, lds, bxi, byi, bzi);  // (two waves per SIMD: registers for the carried plane)
    else
        fine_u4_run<EPI, TX, TY, CARRY, false>(t, a, lds, bxi, byi, bzi);
}

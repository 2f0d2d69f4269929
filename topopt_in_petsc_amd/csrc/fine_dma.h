// fine_dma.h -- third generation of the fine-level matrix-free hex8 operator (matfree_tile.h holds the mathematics,
// fine_tile.h the second generation and the measurements that led here).
//
//   y = (N K(E) N + I - N) u        (LinearElasticity.cc:510-542), fused epilogues as in fine_tile.h
//
// What round 2 measured: at 256^3 the data-movement skeleton of k_fine_tile runs at 0.40 of the HBM peak
// and the time does not follow the bytes (z-chunks of 16 or 129 planes: the same 328 us).  Each workgroup has exactly
// ONE step of loads in flight (4 x 8 B per lane held in VGPRs): a step costs one loaded memory round trip, whatever it
// computes.  More bytes in flight need either registers (there are none left at 3 waves per SIMD) or LDS:
//
//  * node planes and the moduli reach LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`: 16 B per lane, no VGPR, no
//    ds_write), D steps ahead, into rings of D + 2 (planes) and D + 1 (moduli) slots.  A tile row is ONE contiguous run
//    of 16-byte units that starts at the 16-byte boundary at or below its first node: the units of a wave instruction
//    land lane-linear, so the LDS image of a row is [parity pad][nodes ...] and the reader adds the row's parity.
//  * every vector-memory instruction of the loop is issued by hand (inline asm) in a fixed order and count per step, and
//    the waits are counted by hand: the compiler's own s_waitcnt insertion does not know LDS-DMA and would drain the
//    queue at the first ordinary load (cdna_hip_programming.md, "Pipelining across barriers").  Order per step:
//        [top]  batch(s) = planes' units, moduli units          (needed by step s + D)
//        [mid]  s_waitcnt vmcnt((D-1) * OPS) -> batch(s+1-D) has landed for THIS wave; the step's barrier publishes it
//        [end]  epilogue operands of step s+1 (b, u-: VGPR loads), then the stores of step s
//    The prologue issues the same pattern (stores and operand loads against empty descriptors: counted, no traffic) so
//    that one immediate fits every step.  Out-of-range planes and everything beyond the last plane a chunk needs are
//    requested through an EMPTY descriptor: zeros in LDS, no memory traffic (the second generation fetched two planes
//    per chunk that nobody read).
//  * the tile shape is a template parameter: TX lanes along x (16: DPP row shift, 32/64: DPP wave shift), TY rows.
//
// Per-node arithmetic and summation order are those of k_fine_tile: results are bitwise identical to it.
#pragma once
#include <type_traits>

#include "fine_tile.h"

#ifndef FD_ABL
#define FD_ABL 0  // ablation builds: 9 = data movement only (DMA, waits, barrier, stores; no arithmetic), 8 = no memory traffic (every descriptor empty), 7 = no stores
#endif
#ifndef FD_POL
#define FD_POL 0  // experiment: which younger operations may retire before an older load.  0: none (in-order counter), 1: stores, 2: anything
#endif

template <int TX, int TY, int D>
struct FineDma {
    static constexpr int NT = TX * TY, NW = NT / 64;
    static constexpr int TOX = TX - 1, TOY = TY - 1;          // node columns / rows produced per tile
    static constexpr int UROWS = TY + 1, UPTS = TX + 1;        // staged node rows / nodes per row
    static constexpr int UPR_U = (3 * UPTS + 2) / 2;           // 16-byte units per staged node row (parity pad included)
    static constexpr int ROWW_U = 2 * UPR_U;                   // doubles per LDS row
    static constexpr int NUNIT_U = UROWS * UPR_U;
    static constexpr int NIU = (NUNIT_U + 63) / 64;            // wave instructions per plane
    static constexpr int NIU_W = (NIU + NW - 1) / NW;          // ... per wave
    static constexpr int USLOT = NIU * 1024;                   // bytes per ring slot
    static constexpr int UPR_E = (TX + 2) / 2;                 // units per staged modulus row
    static constexpr int ROWW_E = 2 * UPR_E;
    static constexpr int NUNIT_E = TY * UPR_E;
    static constexpr int NIE = (NUNIT_E + 63) / 64;
    static constexpr int NIE_W = (NIE + NW - 1) / NW;
    static constexpr int ESLOT = NIE * 1024;
    static constexpr int NI = NIU_W + NIE_W;                   // DMA instructions per wave and step
    static constexpr int RU = D + 2, RE = D + 1;               // ring depths
    static constexpr int MSZ = ((UROWS * UPTS + 15) / 16) * 16;  // mask bytes per plane (Dirichlet tiles)
    // LDS map (bytes)
    static constexpr int OFF_U = 0;
    static constexpr int OFF_E = OFF_U + RU * USLOT;
    static constexpr int OFF_Y = OFF_E + RE * ESLOT;
    static constexpr int YBUF = (NT - TX) * 3 * 8;             // y-combination buffer: the last row of a tile has no reader
    static constexpr int OFF_M = OFF_Y + 2 * YBUF;
    static constexpr int OFF_RED = OFF_Y;                      // block reduction of the dot epilogues (after the loop)
    static constexpr int LDS_BYTES = OFF_M + 4 * MSZ;
    static_assert(NT % 64 == 0, "whole waves");
    static_assert(TX == 16 || TX == 32 || TX == 64, "rows are 16, 32 or 64 lanes");
};

// left neighbour's value within the tile row (lanes with tx == 0 receive a value that is never used)
template <int TX>
__device__ __forceinline__ double dpp_left(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    if (TX == 16) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xF, 0xF, true);  // row_shr:1
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xF, 0xF, true);
    } else {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xF, 0xF, true);  // wave_shr:1
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xF, 0xF, true);
    }
    return __hiloint2double(hi, lo);
}

typedef double fd_d2 __attribute__((ext_vector_type(2)));
typedef unsigned fd_u4 __attribute__((ext_vector_type(4)));
typedef unsigned fd_u2 __attribute__((ext_vector_type(2)));
constexpr unsigned FD_RSRC_FLAGS = 0x00020000u;
constexpr unsigned FD_OOB = 0x80000000u;  // a voffset no descriptor of this file reaches (arrays < 2 GB, checked by the host)

// Hazards the compiler's recogniser does not see inside inline asm: an SGPR written by the SALU (or by v_readlane /
// v_readfirstlane: spilled descriptor words come back that way) needs 5 wait states before a vector-memory instruction
// reads it, otherwise the instruction may still see the OLD value -- a descriptor of the previous plane.  Every asm
// statement of this file that reads a descriptor therefore opens with its own wait states (the two s_mov + s_nop 2 of
// the DMA form, s_nop 4 elsewhere); the same s_nop covers the M0 write -> LDS-DMA hazard.
// one LDS-DMA wave instruction: 64 lanes x 16 B from rs[voff] to lds_dst + 16 * lane
__device__ __forceinline__ void fd_dma16(unsigned lds_dst, unsigned voff, __amdgpu_buffer_rsrc_t rs) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_dst), "v"(voff), "s"(rs)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void fd_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// voffset of a unit: stored value (rem = 0 offset, bit 0 = "the row's first double is odd") -> offset for this plane's rem.
// Offsets in front of the window (a tile's column -1 in the first row of a plane) are negative: every value >= 2^31 becomes
// FD_OOB exactly, which no descriptor of this file reaches (num_records <= 0x7FFFFF00) -> the hardware returns zeros.
__device__ __forceinline__ unsigned fd_voff(unsigned stored, int rem) {
    const unsigned v = (stored & ~1u) + (((stored & 1u) & (unsigned)rem) << 4);
    return v < FD_OOB ? v : FD_OOB;
}

// descriptor of the 16-byte-aligned window that starts at or below p; *rem = doubles between the window and p (0 | 1)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fd_window(const double *p, const double *end, bool on, int *rem) {
    const unsigned long a = (unsigned long)p;
    *rem = (int)((a >> 3) & 1ul);
    const unsigned long base = a & ~15ul;
    const long room = (long)((unsigned long)end - base);
    const int n = on ? (int)(room > 0x7FFFFF00l ? 0x7FFFFF00l : room) : 0;
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, n, FD_RSRC_FLAGS);
}

template <int EPI, int TX, int TY, int D, bool CARRY, bool MASKED>
__device__ __forceinline__ void fine_dma_run(const TileArgs &t, const NodeArgs &a, char *lds, int bxi, int byi, int bzi) {
#pragma clang fp contract(off)
    using S = FineDma<TX, TY, D>;
    constexpr bool IS_CHEB = (EPI == EPI_CHEB || EPI == EPI_CHEB_DOT);
    constexpr bool DIAG_FLY = IS_CHEB;
    constexpr bool HAS_B = (EPI == EPI_RESID || IS_CHEB);
    constexpr bool IS_DOT = (EPI == EPI_APPLY_DOT || EPI == EPI_CHEB_DOT);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tx = tid % TX, ty = tid / TX;
    const int bx = bxi * S::TOX, by = byi * S::TOY;
    int kz0, kz1;
    tile_chunk(t, bzi, kz0, kz1);
    const int nsteps = kz1 - kz0 + 2;  // element layers kz0-1 .. kz1; node planes jj = 0 .. nsteps (plane kz0-1+jj)
    const int ei = bx - 1 + tx, ej = by - 1 + ty;
    const bool elem_ok = ei >= 0 && ei < t.ex && ej >= 0 && ej < t.ey;
    const bool node_ok = tx >= 1 && ty >= 1 && ei < t.nx && ej < t.ny;
    const long plane = (long)t.nx * t.ny;
    const long lay = (long)t.ex * t.ey;
    const long ncol = node_ok ? (long)ei + (long)t.nx * ej : 0;
    const int yprev = ty >= 1 ? tid - TX : tid;
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char *)lds;

    // ---- DMA units of this lane.  Unit w of an array: row r = w / UPR, 16-byte column q = w % UPR; its byte offset
    // inside the plane's aligned window is 8 * ((rem + c_r) & ~1) + 16 q, c_r = first double of the row relative to the
    // plane.  Stored as the rem = 0 value with bit 0 = "c_r is odd" (then rem = 1 moves the unit up by 16 bytes).
    unsigned vU[S::NIU_W], vE[S::NIE_W];
    unsigned dstU[S::NIU_W], dstE[S::NIE_W];  // LDS byte offset inside the slot (wave-uniform)
#pragma unroll
    for (int k = 0; k < S::NIU_W; k++) {
        // wave instruction of the plane; the instructions that pad the per-wave count repeat the last one (same
        // units to the same place: no traffic beyond the L1, and every wave issues the same number per step)
        const int j = min(k * S::NW + wave, S::NIU - 1);
        const int w = j * 64 + lane;
        dstU[k] = (unsigned)(j * 1024);
        const int r = w / S::UPR_U, q = w % S::UPR_U;
        const int gj = min(max(by - 1 + r, 0), t.ny - 1);
        const int c = 3 * (bx - 1 + t.nx * gj);
        const int v0 = 8 * (c & ~1) + 16 * q;
        vU[k] = w < S::NUNIT_U ? ((unsigned)v0 | (unsigned)(c & 1)) : FD_OOB;  // (negative: clamped to FD_OOB per step)
    }
#pragma unroll
    for (int k = 0; k < S::NIE_W; k++) {
        const int j = min(k * S::NW + wave, S::NIE - 1);
        const int w = j * 64 + lane;
        dstE[k] = (unsigned)(j * 1024);
        const int r = w / S::UPR_E, q = w % S::UPR_E;
        const int gj = min(max(by - 1 + r, 0), t.ey - 1);
        const int c = bx - 1 + t.ex * gj;
        const int v0 = 8 * (c & ~1) + 16 * q;
        vE[k] = w < S::NUNIT_E ? ((unsigned)v0 | (unsigned)(c & 1)) : FD_OOB;
    }
    // rows of this thread: parity of the row's first double relative to the plane
    // (rows clamped at the domain boundary only feed elements outside the domain, whose modulus is taken as 0: their
    // parity -- any finite double will do -- is not tracked)
    const int oddU0 = (bx - 1 + t.nx * ej) & 1, oddU1 = oddU0 ^ (t.nx & 1);
    const int oddE = (bx - 1 + t.ex * ej) & 1, oddEd = oddE ^ (t.ex & 1);
    const unsigned aU0 = (unsigned)((ty * S::ROWW_U + 3 * tx) * 8), aU1 = aU0 + S::ROWW_U * 8;
    (void)aU1;
    const unsigned aE = (unsigned)((ty * S::ROWW_E + tx) * 8);

    const double *xend = a.x + 3 * plane * t.nzl, *Eend = t.E + lay * t.ezl;
    // window remainder (0 | 1 doubles) of node plane jj / modulus layer jl of this chunk; wave-uniform scalar arithmetic
    auto rem_u = [&](int jj) -> int {
        const int pc = min(max(kz0 - 1 + jj, 0), t.nzl - 1);
        return (int)(((unsigned long)(a.x + 3 * plane * pc) >> 3) & 1ul);
    };
    auto rem_e = [&](int jl) -> int {
        const int lc = min(max(kz0 - 1 + jl, 0), t.ezl - 1);
        return (int)(((unsigned long)(t.E + lay * lc) >> 3) & 1ul);
    };
    // batch(tb): plane jj = tb + 1 + D into slot jj % RU, moduli of layer jl = tb + D into slot jl % RE
    auto batch = [&](int tb) {
        const int jj = tb + 1 + D, jl = tb + D;
        {
            const int p = kz0 - 1 + jj, pc = min(max(p, 0), t.nzl - 1);
            int rem;
            const __amdgpu_buffer_rsrc_t rs = fd_window(a.x + 3 * plane * pc, xend, p == pc && jj <= nsteps && FD_ABL != 8, &rem);
            const unsigned base = lds0 + S::OFF_U + (jj % S::RU) * S::USLOT;
#pragma unroll
            for (int k = 0; k < S::NIU_W; k++) fd_dma16(base + dstU[k], fd_voff(vU[k], rem), rs);
        }
        {
            const int l = kz0 - 1 + jl, lc = min(max(l, 0), t.ezl - 1);
            int rem;
            const __amdgpu_buffer_rsrc_t rs = fd_window(t.E + lay * lc, Eend, l == lc && jl >= 0 && jl < nsteps && FD_ABL != 8, &rem);
            const unsigned base = lds0 + S::OFF_E + ((jl + S::RE) % S::RE) * S::ESLOT;
#pragma unroll
            for (int k = 0; k < S::NIE_W; k++) fd_dma16(base + dstE[k], fd_voff(vE[k], rem), rs);
        }
    };

    // ---- epilogue operands (b, u-) and the stores: one voffset per lane inside an output plane
    const unsigned voff_out = node_ok ? 24u * (unsigned)ncol : 0x7FFFFF00u;
    const bool read_prev = IS_CHEB && a.c1 != 0.0 && !a.prev_zero;  // uniform, loop invariant
    fd_d2 b01 = {0.0, 0.0}, p01 = {0.0, 0.0};
    double b2 = 0.0, p2 = 0.0;
    constexpr int NLE = HAS_B ? (IS_CHEB ? 4 : 2) : 0;  // operand load instructions per step (u- loads are issued in both variants)
    constexpr int OPS = S::NI + NLE + 2;
    // wait immediates: operations issued after the awaited one that are ASSUMED still outstanding when it completes
    constexpr int W_MID = FD_POL == 0 ? (D - 1) * OPS : (FD_POL == 1 ? (D - 1) * (S::NI + NLE) : 0);
    constexpr int W_PRO = FD_POL == 0 ? (NLE + 2) + (D - 1) * OPS : (FD_POL == 1 ? NLE + (D - 1) * (S::NI + NLE) : 0);
    constexpr int W_EPI = FD_POL == 0 ? 2 + S::NI : (FD_POL == 1 ? S::NI : 0);
    auto tail_loads = [&](int pl, bool on) {  // operands of output plane pl
        if (!HAS_B) return;
        if (FD_ABL == 8) on = false;
        const int pc = min(max(pl, 0), t.nzl - 1);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(a.b) + 3 * plane * pc, 0, on ? (int)(24 * plane) : 0, FD_RSRC_FLAGS);
        // ("+v": the registers stay b01 / b2 from the first statement to the last wait -- a load that lands in a register the
        // compiler has meanwhile given to something else corrupts it silently)
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, 0 offen\n\tbuffer_load_dwordx2 %1, %2, %3, 0 offen offset:16"
                     : "+v"(b01), "+v"(b2)
                     : "v"(voff_out), "s"(rb)
                     : "memory");
        if (IS_CHEB) {
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(a.out + 3 * plane * pc, 0, (on && read_prev) ? (int)(24 * plane) : 0, FD_RSRC_FLAGS);
            asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, 0 offen\n\tbuffer_load_dwordx2 %1, %2, %3, 0 offen offset:16"
                         : "+v"(p01), "+v"(p2)
                         : "v"(voff_out), "s"(rp)
                         : "memory");
        }
    };
    auto tail_stores = [&](int pl, bool on, const double o[3]) {
        if (FD_ABL == 8 || FD_ABL == 7) on = false;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.out + 3 * plane * max(pl, 0), 0, on ? (int)(24 * plane) : 0, FD_RSRC_FLAGS);
        const fd_d2 o01 = {o[0], o[1]};
        asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %2, %3, 0 offen\n\tbuffer_store_dwordx2 %1, %2, %3, 0 offen offset:16\n\ts_nop 1"  // (+ store wider than 8 bytes -> overwrite of its data registers)
                     :
                     : "v"(o01), "v"(o[2]), "v"(voff_out), "s"(rs)
                     : "memory");
    };

    // ---- Dirichlet tiles: mask bytes of the staged nodes of a plane (ordinary loads one step ahead, as in fine_tile.h;
    // the compiler's waits for them are conservative with respect to the hand-counted queue)
    constexpr int NMK = (S::UROWS * S::UPTS + S::NT - 1) / S::NT;
    unsigned mk_off[NMK];
#pragma unroll
    for (int s = 0; s < NMK; s++) {
        const int n = min(tid + s * S::NT, S::UROWS * S::UPTS - 1);
        const int gi = min(max(bx - 1 + n % S::UPTS, 0), t.nx - 1), gj = min(max(by - 1 + n / S::UPTS, 0), t.ny - 1);
        mk_off[s] = (unsigned)(gi + t.nx * gj);
    }
    auto load_mask = [&](int p, unsigned mk[NMK]) {
        const uint8_t *__restrict__ mp = t.mask + plane * min(max(p, 0), t.nzl - 1);
#pragma unroll
        for (int s = 0; s < NMK; s++) mk[s] = mp[mk_off[s]];
    };
    auto slot_mask = [&](int jj) -> uint8_t * { return (uint8_t *)(lds + S::OFF_M + (jj & 3) * S::MSZ); };
    auto store_mask = [&](int jj, const unsigned mk[NMK]) {
        uint8_t *m = slot_mask(jj);
#pragma unroll
        for (int s = 0; s < NMK; s++) m[min(tid + s * S::NT, S::UROWS * S::UPTS - 1)] = (uint8_t)mk[s];
    };
    const int n00 = ty * S::UPTS + tx;

    // The element's 8 nodes in the Walsh-Hadamard basis: planes jj (bottom) and jj + 1 (top) of the ring.
    // CARRY = false: both planes are read every step -- carrying the transformed bottom plane over from the previous step
    //   (k_fine_tile) costs 24 VGPRs, re-reading it 12 ds_read_b64 and 24 additions; with the registers the Chebyshev
    //   epilogue fits 3 waves per SIMD without spilling (a spill reload is a vector-memory operation: its wait would drain
    //   the DMA queue) and the plain product 4.
    // CARRY = true: the transformed top plane of a step is the bottom plane of the next (Ub).
    // xo = own node of the bottom plane, unmasked.
    double Ub[3][4];
    constexpr bool NEED_XO = IS_CHEB || EPI == EPI_APPLY_DOT || MASKED;
    auto read_plane = [&](int jj, double U[3][4], double x0[3]) {  // raw own node in x0, masked + transformed plane in U
        const int rem = rem_u(jj);
        const char *sp = lds + S::OFF_U + (jj % S::RU) * S::USLOT;
        const double *r0 = (const double *)(sp + aU0 + 8 * (rem ^ oddU0)), *r1 = (const double *)(sp + aU0 + S::ROWW_U * 8 + 8 * (rem ^ oddU1));
        unsigned mk[4] = {0, 0, 0, 0};
        if (MASKED) {
            const uint8_t *m = slot_mask(jj);
            mk[0] = m[n00], mk[1] = m[n00 + 1], mk[2] = m[n00 + S::UPTS], mk[3] = m[n00 + S::UPTS + 1];
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            U[c][0] = r0[c], U[c][1] = r0[3 + c], U[c][2] = r1[c], U[c][3] = r1[3 + c];
            x0[c] = U[c][0];
            if (MASKED) {
#pragma unroll
                for (int q = 0; q < 4; q++) U[c][q] = ((mk[q] >> c) & 1u) ? 0.0 : U[c][q];
            }
            wht4(U[c]);
        }
    };
    auto read_element = [&](int jj, double u[3][8], double xo[3]) {
        double Ut[3][4], xt[3];
        if (!CARRY) {
            read_plane(jj, Ub, xo);
        } else if (NEED_XO) {
            const double *r0 = (const double *)(lds + S::OFF_U + (jj % S::RU) * S::USLOT + aU0 + 8 * (rem_u(jj) ^ oddU0));
#pragma unroll
            for (int c = 0; c < 3; c++) xo[c] = r0[c];
        }
        read_plane(jj + 1, Ut, xt);
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                u[c][m] = Ub[c][m] + Ut[c][m];
                u[c][m + 4] = Ub[c][m] - Ut[c][m];
                if (CARRY) Ub[c][m] = Ut[c][m];
            }
    };
    auto read_E = [&](int jl) -> double {
        return *(const double *)(lds + S::OFF_E + (jl % S::RE) * S::ESLOT + aE + 8 * (rem_e(jl) ^ oddE));
    };
    // DIAG_FLY: moduli of the left / lower / lower-left neighbour columns of layer jl, straight from the staged rows
    // (k_fine_tile exchanged the in-plane pair sums through LDS: same operands, same order of additions)
    const bool okL = tx >= 1 && ei - 1 >= 0 && ei - 1 < t.ex && ej >= 0 && ej < t.ey;
    const bool okD = ty >= 1 && ei >= 0 && ei < t.ex && ej - 1 >= 0 && ej - 1 < t.ey;
    const bool okDL = tx >= 1 && ty >= 1 && ei - 1 >= 0 && ei - 1 < t.ex && ej - 1 >= 0 && ej - 1 < t.ey;
    // (lanes of the first column / row read the doubles in front of their row / slot: LDS of this workgroup, unused)
    auto read_E4 = [&](int jl, double Ee) -> double {
        const char *sp = lds + S::OFF_E + (jl % S::RE) * S::ESLOT + aE;
        const int rem = rem_e(jl);
        const double *own = (const double *)(sp + 8 * (rem ^ oddE)), *dn = (const double *)(sp - S::ROWW_E * 8 + 8 * (rem ^ oddEd));
        const double eL = own[-1], eD = dn[0], eDL = dn[-1];
        const double ex2 = Ee + (okL ? eL : 0.0);
        return ex2 + ((okD ? eD : 0.0) + (okDL ? eDL : 0.0));
    };
    double(*s_y)[(S::NT - TX) * 3] = (double(*)[(S::NT - TX) * 3])(lds + S::OFF_Y);

    // ---- prologue: batches -D-1 .. -1 with the queue pattern of a step each
    unsigned pmk[NMK];
    const double zero3[3] = {0.0, 0.0, 0.0};
    // (the operand loads of the pattern are replaced by dropped stores: an operation is an operation for the counter,
    // and a load into a register nobody waits for would land in whatever the compiler keeps there by then)
    for (int tb = -D - 1; tb <= -1; tb++) {
        batch(tb);
#pragma unroll
        for (int q = 0; q < NLE / 2 + 1; q++) tail_stores(kz0 - 1, false, zero3);
    }
    if (MASKED) {
        unsigned m0[NMK], m1[NMK];
        load_mask(kz0 - 1, m0);
        load_mask(kz0, m1);
        load_mask(kz0 + 1, pmk);
        store_mask(0, m0);
        store_mask(1, m1);
    }
    fd_wait<W_PRO>();  // batches -D-1 and -D have landed: planes 0, 1, moduli of layer 0
    __syncthreads();
    // a wave whose lanes hold neither an element nor a node of the domain (rows beyond the last node row of the mesh)
    // only takes part in the staging, the barriers and the (dropped) memory operations: idle_step below.  (The row
    // below the first node row, ej < 0, is NOT idle: the first node row reads its -- zero -- contribution.)
    const bool live = __builtin_amdgcn_readfirstlane((int)(__ballot(elem_ok || ej < 0 || (tx >= 1 && ty >= 1 && ei < t.nx && ej < t.ny)) != 0ull)) != 0;
    if (CARRY && live) {
        double x0[3];
        read_plane(0, Ub, x0);
    }
    double Cy[3][4];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int m = 0; m < 4; m++) Cy[c][m] = 0.0;
    double pdot = 0.0;
    double Elow = 0.0;

    auto step = [&](int s, auto with_prev) {
        constexpr bool PREV = decltype(with_prev)::value;
        const int el = kz0 - 1 + s;
        batch(s);
        if (MASKED) {
            store_mask(s + 2, pmk);
            load_mask(el + 3, pmk);
        }
        if (FD_ABL == 9) {  // timing skeleton: the step's memory operations, waits and barrier; no arithmetic
            double xo9[3], u9[3][8];
            read_element(s, u9, xo9);
#pragma unroll
            for (int c = 0; c < 3; c++) xo9[c] += u9[c][0] + u9[c][7];
            const double e9 = read_E(s);
            fd_wait<W_MID>();
            __syncthreads();
            if (HAS_B) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(b01), "+v"(b2), "+v"(p01), "+v"(p2) : "n"(W_EPI));
            double o9[3] = {xo9[0] + b01.x + p01.x, xo9[1] + b01.y + p01.y, xo9[2] + b2 + p2 + e9};
#pragma unroll
            for (int c = 0; c < 3; c++) asm volatile("" : "+v"(o9[c]));
            tail_loads(el + 1, s + 1 < nsteps);
            tail_stores(el, s >= 1, o9);
            return;
        }
        const double Eraw = read_E(s);
        // ---- element in the Walsh-Hadamard basis
        double xo[3], u[3][8], f[3][8];
        read_element(s, u, xo);
        int boff;
        asm volatile("s_mov_b32 %0, %1" : "=s"(boff) : "s"(t.slot_off));
        sym_ke_blocks(c_symB + boff, u, f);
        sym_ke_translation<KrylovEpi<EPI>::value>(c_symX + 4 * boff, u, f);
        const double Ee = elem_ok ? Eraw : 0.0;  // (k_fine_tile multiplies by 1.0 / 0.0: the same bits)
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
            s0[c] = P[c][0] + dpp_left<TX>(P[c][1]);
            const double up = P[c][2] + dpp_left<TX>(P[c][3]);
            if (ty < TY - 1) s_y[s & 1][tid * 3 + c] = up;
        }
        double e4 = 0.0;
        if (DIAG_FLY) e4 = read_E4(s, Ee);
        unsigned mown = 0;
        if (MASKED) mown = slot_mask(s)[n00];
        fd_wait<W_MID>();  // this wave's part of batch(s + 1 - D): published by the barrier
        __syncthreads();
        // ---- epilogue
        double di[3] = {0, 0, 0};
        if (DIAG_FLY) {
            const double rinv = 1.0 / (e4 + Elow);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                di[c] = rinv * c_symB[boff + SYMKE_N + c];
                if (MASKED) di[c] = ((mown >> c) & 1u) ? 1.0 : di[c];
            }
        }
        if (HAS_B) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(b01), "+v"(b2), "+v"(p01), "+v"(p2) : "n"(W_EPI));
        const double bo[3] = {b01.x, b01.y, b2}, dd[3] = {p01.x, p01.y, p2};
        double o[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            double y = s0[c] + s_y[s & 1][yprev * 3 + c];
            if (MASKED) y = ((mown >> c) & 1u) ? xo[c] : y;
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
        if (DIAG_FLY) Elow = e4;
#pragma unroll
        for (int c = 0; c < 3; c++) asm volatile("" : "+v"(o[c]));
        tail_loads(el + 1, s + 1 < nsteps);
        tail_stores(el, s >= 1, o);
    };
    // the same queue pattern and barriers without the element: waves that hold no part of the domain
    auto idle_step = [&](int s) {
        batch(s);
        if (MASKED) {
            store_mask(s + 2, pmk);
            load_mask(kz0 - 1 + s + 3, pmk);
        }
        fd_wait<W_MID>();
        __syncthreads();
        if (HAS_B) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(b01), "+v"(b2), "+v"(p01), "+v"(p2) : "n"(W_EPI));
        tail_loads(kz0 + s, false);
        tail_stores(kz0 - 1 + s, false, zero3);
    };
    if (!live) {
        for (int s = 0; s < nsteps; s++) idle_step(s);
    } else if (read_prev) {
        for (int s = 0; s < nsteps; s++) step(s, std::true_type{});
    } else {
        for (int s = 0; s < nsteps; s++) step(s, std::false_type{});
    }
    // no LDS-DMA may land after the workgroup has given its LDS back, no operand load in a register that is somebody else's
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(b01), "+v"(b2), "+v"(p01), "+v"(p2)::"memory");
    if (MASKED) {
        // The last steps requested mask bytes nobody reads.  Use them, so that the compiler waits for them HERE: its
        // wait-count analysis follows a (never taken) structured edge from this variant into the unmasked one and would
        // otherwise protect "their" registers there with a vmcnt(0) inside the hand-counted loop.
#pragma unroll
        for (int q = 0; q < NMK; q++) asm volatile("" ::"v"(pmk[q]));
    }
    if (IS_DOT) {
        // block total in thread 0 (any workgroup size)
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

template <int EPI, int TX, int TY, int D, int WPS, bool CARRY = false>
__global__ __launch_bounds__(TX *TY, WPS) void k_fine_dma(TileArgs t, NodeArgs a) {
    using S = FineDma<TX, TY, D>;
    __shared__ __attribute__((aligned(1024))) char lds[S::LDS_BYTES];
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
    bool masked = false;
    if (t.colmask) {
        unsigned any = 0;
        for (int n = threadIdx.x; n < S::UROWS * S::UPTS; n += S::NT) {
            const int gi = min(max(bxi * S::TOX - 1 + n % S::UPTS, 0), t.nx - 1), gj = min(max(byi * S::TOY - 1 + n / S::UPTS, 0), t.ny - 1);
            any |= t.colmask[gi + t.nx * gj];
        }
        // (uniform, and the compiler must know it: behind a divergent branch the two variants share one structured
        // control flow, and the mask loads still pending at the end of the first reach the wait-count analysis of the
        // second -- it then puts a vmcnt(0) into the hand-counted loop)
        masked = __builtin_amdgcn_readfirstlane(__syncthreads_or(any != 0u)) != 0;
    }
    if (masked)
        fine_dma_run<EPI, TX, TY, D, CARRY, true>(t, a, lds, bxi, byi, bzi);
    else
        fine_dma_run<EPI, TX, TY, D, CARRY, false>(t, a, lds, bxi, byi, bzi);
}

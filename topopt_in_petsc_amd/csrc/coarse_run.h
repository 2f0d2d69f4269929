// coarse_run.h -- the Chebyshev run of the coarsest level (30-60 steps on a few hundred to a few thousand nodes) in ONE
// launch.  As separate launches every step costs the dependent-dispatch latency of the device (~6.5 us for a ~2 us
// kernel; measured, also under hipGraph replay), 58 % of all launches of a design iteration at 128^3.  Here the steps
// are iterations of a loop inside one kernel of at most 64 resident workgroups:
//   * every thread keeps ITS part of the stencil in registers for the whole run (a row is split over 9 threads by the
//     (z, y) offset of the neighbour, exactly as in k_dia_row_split<.,.,9>; R rows per thread), the part-0 thread of a
//     row also keeps the row's iterate, direction, right-hand side and Jacobi factor;
//   * per step only the iterate travels: written with agent-scope stores into one of two buffers; after a barrier over
//     the workgroups (one arrival counter, release / relaxed polling: 1.4 us for 8, 2.3-3.3 us for 64 workgroups --
//     tools/probe/barrier_probe.hip) every workgroup copies the contiguous stretch of the iterate its rows couple to
//     (own nodes +- one plane, one row, one node) into LDS with ONE round of coalesced cache-bypassing loads (sc1: the
//     L2 slices of the XCDs are not coherent with each other).  First version: agent-scope ATOMIC loads straight from
//     the stencil loop -- the compiler waits for each of them (s_waitcnt vmcnt(0) after every load), 36 serial round
//     trips per step: 11 us per step, slower than the launches;
//   * the arithmetic of a row is the one of k_dia_row_split<DOF, EPI_CHEB, 9>: same products, same order, same bits.
//   * a level of at most 448 rows (coarsest grids of <= ~150 nodes) is ONE workgroup's work: the iterate then lives in
//     LDS for the whole run, nothing travels and nobody waits -- 0.4-0.5 us per step instead of ~3 (SINGLE).
// A workgroup that waits longer than ~1 s (its peers not resident: cannot happen with <= 64 workgroups on an otherwise
// idle queue, but a hang would take the device down) gives up everywhere and poisons the result with NaN, which the
// Krylov loop reports as divergence.
#pragma once
#include "operators.h"

constexpr int RUN_WG = 512, RUN_RPB = RUN_WG / 9;  // 56 rows x 9 parts per round of a workgroup
constexpr int RUN_MAXK = 96, RUN_MAX_WGS = 64;
constexpr int RUN_XS = 6144;  // doubles of the iterate a workgroup stages per step (48 KB)

struct ChebRunCoef {
    double c1[RUN_MAXK], c2[RUN_MAXK];
    int nsteps;
};

// doubles of the iterate that a workgroup of R x RUN_RPB rows couples to (upper bound; host-side eligibility check)
inline long run_stage_doubles(const Geom &g, int dof, int R) {
    const long own = ((long)RUN_RPB * R + dof - 1) / dof + 1;
    return dof * (own + 2 * (g.plane() + g.nx + 1));
}

template <int DOF, int R, bool SINGLE>
__global__ __launch_bounds__(RUN_WG) void k_dia_cheb_run(DiaOp<DOF> op, const double *__restrict__ b,
                                                         const double *__restrict__ dinv, const double *__restrict__ d0,
                                                         double *xa, double *xb, ChebRunCoef cr, unsigned long long *cnt,
                                                         unsigned long long base) {
    __shared__ double s_part[9][RUN_RPB * R];
    __shared__ double xs[RUN_XS];
    __shared__ int s_dead;
    const Geom &g = op.g;
    const long plane = g.plane();
    const long nown = g.owned_nodes() * DOF;
    const long off = plane * g.own_lo * DOF;  // first owned row
    const int part = threadIdx.x / RUN_RPB, r = threadIdx.x % RUN_RPB;
    const bool lane_ok = part < 9;
    // the stretch of nodes this workgroup reads: its rows' nodes +- (plane + nx + 1), clamped to the array
    const long t_lo = (long)blockIdx.x * R * RUN_RPB, t_hi = min(t_lo + (long)R * RUN_RPB, nown) - 1;
    const long reach = plane + g.nx + 1;
    const long n_first = max((t_lo + off) / DOF - reach, 0L);
    const long n_last = min((t_hi + off) / DOF + reach, g.nodes() - 1);
    const int stage_n = (int)((n_last - n_first + 1) * DOF);  // <= RUN_XS (checked by the host)
    // ---- stencil parts: thread (part, r) serves rows t_lo + m * RUN_RPB + r, m < R, with the 3 x DOF values of its
    // (z, y) neighbour offset
    double coef[R][3 * DOF];
    int nbi[R][3];
    bool valid[R];
#pragma unroll
    for (int m = 0; m < R; m++) {
        const long t = t_lo + (long)m * RUN_RPB + r;
        valid[m] = lane_ok && t < nown;
        const long q = (valid[m] ? t : 0) + off;
        const long n = q / DOF;
        const int k = (int)(n / plane);
        const int rem = (int)(n % plane);
        const int j = rem / g.nx, i = rem % g.nx;
        const int dk = part / 3 - 1, dj = part % 3 - 1;
        const bool okj = k + dk >= 0 && k + dk < g.nzl && j + dj >= 0 && j + dj < g.ny;
#pragma unroll
        for (int di = -1; di <= 1; di++) {
            const bool ok = okj && i + di >= 0 && i + di < g.nx;
            const int blk = ((lane_ok ? dk : 0) + 1) * 9 + ((lane_ok ? dj : 0) + 1) * 3 + (di + 1);
            // coefficients of non-existent neighbours are stored as zeros: only the address is made safe
            const long nb = ok ? n + di + (long)g.nx * (dj + (long)g.ny * dk) : n;
            nbi[m][di + 1] = valid[m] ? (int)((nb - n_first) * DOF) : 0;  // index into the staged stretch
#pragma unroll
            for (int c = 0; c < DOF; c++) coef[m][(di + 1) * DOF + c] = valid[m] ? op.S[(long)(blk * DOF + c) * op.nrows + q] : 0.0;
        }
    }
    // ---- the rows are FINISHED (partial sums added, Chebyshev update) one per thread: thread f takes row t_lo + f
    const int f = threadIdx.x;
    const bool fin = f < R * RUN_RPB && t_lo + f < nown;
    const long qf = (fin ? t_lo + f : 0) + off;
    const int xsf = (int)(qf - n_first * DOF);  // the row's own place in the staged stretch
    const double e_b = fin ? b[qf] : 0.0, e_di = fin ? dinv[qf] : 0.0;
    double dcur = fin ? d0[qf] : 0.0, xo = fin ? xa[qf] : 0.0;
    const double *xin = xa;
    double *xout = xb;
    bool dead = false;
    if (SINGLE) {  // the whole level, once
        for (int idx = threadIdx.x; idx < stage_n; idx += RUN_WG) xs[idx] = xa[n_first * DOF + idx];
        __syncthreads();
    }
    for (int s = 0; s < cr.nsteps; s++) {
        if (!SINGLE) {
            // ---- this step's input: one round of coalesced loads past the (non-coherent) L2; all loads first (offsets
            // past the stretch are dropped by the buffer bounds check), then the LDS writes
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(xin) + n_first * DOF, 0, stage_n * 8, 0x00020000);
            constexpr int NST = RUN_XS / RUN_WG;
            double tmp[NST];
#pragma unroll
            for (int q = 0; q < NST; q++)
                tmp[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (threadIdx.x + q * RUN_WG) * 8, 0, 16 /* sc1 */));
#pragma unroll
            for (int q = 0; q < NST; q++)
                if (threadIdx.x + q * RUN_WG < stage_n) xs[threadIdx.x + q * RUN_WG] = tmp[q];
            __syncthreads();
        }
#pragma unroll
        for (int m = 0; m < R; m++) {
            double y = 0.0;
            if (valid[m]) {
#pragma unroll
                for (int d3 = 0; d3 < 3; d3++)
#pragma unroll
                    for (int c = 0; c < DOF; c++) y = fma(coef[m][d3 * DOF + c], xs[nbi[m][d3] + c], y);
            }
            if (lane_ok) s_part[part][m * RUN_RPB + r] = y;
        }
        __syncthreads();  // partial sums complete; every read of this step's iterate done
        if (fin) {
            double y = s_part[0][f];
#pragma unroll
            for (int p = 1; p < 9; p++) y += s_part[p][f];
            const double dn = cheb_dn(cr.c1[s], dcur, cr.c2[s], e_di, e_b, y);
            dcur = dn;
            xo = xo + dn;
            if (SINGLE) xs[xsf] = xo;
            else __hip_atomic_store(&xout[qf], xo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (SINGLE) {
            __syncthreads();
            continue;
        }
        if (s + 1 == cr.nsteps) break;  // nobody reads this kernel's last output before the kernel ends
        // ---- barrier over the workgroups of the run
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long target = base + (unsigned long long)(s + 1) * gridDim.x;
            long spins = 0;
            int gave_up = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                // (the give-up flag is looked at every 4096 polls only: a second load per poll doubles the wake-up latency)
                if ((++spins & 4095) == 0 &&
                    (spins > 2000000L || __hip_atomic_load(cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __hip_atomic_store(cnt + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gave_up = 1;
                    break;
                }
            }
            s_dead = gave_up;
        }
        __syncthreads();
        if (s_dead) {
            dead = true;
            break;
        }
        const double *tmp = xout;
        xout = const_cast<double *>(xin);
        xin = tmp;
    }
    if (SINGLE && fin) xa[qf] = xo;  // the result goes back to where the start came from
    if (dead && fin) xa[qf] = xb[qf] = __builtin_nan("");
}


// ---------------------------------------------------------------------------------------------------------------------
// The same run with all its workgroups on ONE XCD (round 3).  The step above is slow because the XCDs' L2 slices are not
// coherent: the iterate goes out with write-through stores and comes back past the L2 (2.6-3.6 us per step, the price of
// a launch).  Inside one XCD the L2 IS the common memory: plain stores land there, L1-bypassing loads (sc1) and the
// arrival counter are served from there -- tools/probe/xcd_probe.hip: 1.7-1.9 us per step for 8-32 workgroups that
// exchange the whole iterate, against 3.2-3.6 us for workgroups anywhere and 3.0 us per launch.
// Placement is not something HIP promises, so it is established at run time: 8 x P workgroups are launched, each reads
// the id of the XCD it runs on (HW_REG_XCC_ID) and takes a ticket on that XCD's join counter; the first XCD to hand out
// P tickets wins (8 P workgroups on 8 XCDs: at least one does), its first P ticket holders are ranks 0 .. P-1 of the
// run, everybody else leaves at once.  Which rows a workgroup serves depends on its ticket, the arithmetic of a row
// does not: same bits as the launches.  The control block is left zeroed by the last workgroup of the 8 P to finish.
// Give-up: a workgroup that waits ~1 s (join or barrier) raises the flag, the run poisons its result with NaN and the
// Krylov loop reports divergence (the host then resets the block); TP_NO_COARSE_XCD=1 switches the path off.
struct XcdRunCtrl {
    unsigned long long join[8][16];  // one cache line each
    unsigned long long winner[16];   // 0: none yet, else 1 + XCC id
    unsigned long long cnt[16];      // barrier arrivals of this run
    unsigned long long finished[16]; // workgroups of the launch that have left
    unsigned long long gaveup[16];   // sticky until the host clears it
};

__device__ inline unsigned run_xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

template <int DOF, int R>
__global__ __launch_bounds__(RUN_WG) void k_dia_cheb_run_xcd(DiaOp<DOF> op, const double *__restrict__ b, const double *__restrict__ dinv,
                                                             const double *__restrict__ d0, double *xa, double *xb, ChebRunCoef cr,
                                                             XcdRunCtrl *ctl, int P) {
    __shared__ double s_part[9][RUN_RPB * R];
    __shared__ double xs[RUN_XS];
    __shared__ int s_dead, s_rank;
    // ---- who takes part
    if (threadIdx.x == 0) {
        const unsigned xcc = run_xcc_id();
        int rank = -1;
        const unsigned long long tk = __hip_atomic_fetch_add(&ctl->join[xcc][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tk < (unsigned long long)P) {
            if (tk == (unsigned long long)(P - 1)) {
                unsigned long long expect = 0ull;
                __hip_atomic_compare_exchange_strong(&ctl->winner[0], &expect, 1ull + xcc, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            unsigned long long w;
            long spins = 0;
            while ((w = __hip_atomic_load(&ctl->winner[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0ull) {
                if (++spins > 2000000L || ((spins & 1023) == 0 && __hip_atomic_load(&ctl->gaveup[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __hip_atomic_store(&ctl->gaveup[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            rank = (w == 1ull + xcc) ? (int)tk : -1;
        }
        s_rank = rank;
    }
    __syncthreads();
    const int rank = s_rank;
    auto leave = [&]() {  // the last of the 8 P workgroups to leave hands the control block back zeroed
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long f = __hip_atomic_fetch_add(&ctl->finished[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (f + 1 == (unsigned long long)gridDim.x) {
                for (int x = 0; x < 8; x++) __hip_atomic_store(&ctl->join[x][0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ctl->winner[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ctl->cnt[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ctl->finished[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    if (rank < 0) {
        leave();
        return;
    }
    const Geom &g = op.g;
    const long plane = g.plane();
    const long nown = g.owned_nodes() * DOF;
    const long off = plane * g.own_lo * DOF;
    const int part = threadIdx.x / RUN_RPB, r = threadIdx.x % RUN_RPB;
    const bool lane_ok = part < 9;
    const long t_lo = (long)rank * R * RUN_RPB, t_hi = min(t_lo + (long)R * RUN_RPB, nown) - 1;
    const long reach = plane + g.nx + 1;
    const long n_first = max((t_lo + off) / DOF - reach, 0L);
    const long n_last = min((t_hi + off) / DOF + reach, g.nodes() - 1);
    const int stage_n = (int)((n_last - n_first + 1) * DOF);
    double coef[R][3 * DOF];
    int nbi[R][3];
    bool valid[R];
#pragma unroll
    for (int m = 0; m < R; m++) {
        const long t = t_lo + (long)m * RUN_RPB + r;
        valid[m] = lane_ok && t < nown;
        const long q = (valid[m] ? t : 0) + off;
        const long n = q / DOF;
        const int k = (int)(n / plane);
        const int rem = (int)(n % plane);
        const int j = rem / g.nx, i = rem % g.nx;
        const int dk = part / 3 - 1, dj = part % 3 - 1;
        const bool okj = k + dk >= 0 && k + dk < g.nzl && j + dj >= 0 && j + dj < g.ny;
#pragma unroll
        for (int di = -1; di <= 1; di++) {
            const bool ok = okj && i + di >= 0 && i + di < g.nx;
            const int blk = ((lane_ok ? dk : 0) + 1) * 9 + ((lane_ok ? dj : 0) + 1) * 3 + (di + 1);
            const long nb = ok ? n + di + (long)g.nx * (dj + (long)g.ny * dk) : n;
            nbi[m][di + 1] = valid[m] ? (int)((nb - n_first) * DOF) : 0;
#pragma unroll
            for (int c = 0; c < DOF; c++) coef[m][(di + 1) * DOF + c] = valid[m] ? op.S[(long)(blk * DOF + c) * op.nrows + q] : 0.0;
        }
    }
    const int f = threadIdx.x;
    const bool fin = f < R * RUN_RPB && t_lo + f < nown;
    const long qf = (fin ? t_lo + f : 0) + off;
    const double e_b = fin ? b[qf] : 0.0, e_di = fin ? dinv[qf] : 0.0;
    double dcur = fin ? d0[qf] : 0.0, xo = fin ? xa[qf] : 0.0;
    const double *xin = xa;
    double *xout = xb;
    bool dead = false;
    for (int s = 0; s < cr.nsteps; s++) {
        {   // this step's input: the stretch of the iterate my rows couple to, past the L1 (served by the XCD's L2)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(xin) + n_first * DOF, 0, stage_n * 8, 0x00020000);
            constexpr int NST = RUN_XS / RUN_WG;
            double tmp[NST];
#pragma unroll
            for (int q = 0; q < NST; q++)
                tmp[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (threadIdx.x + q * RUN_WG) * 8, 0, 16 /* sc1 */));
#pragma unroll
            for (int q = 0; q < NST; q++)
                if (threadIdx.x + q * RUN_WG < stage_n) xs[threadIdx.x + q * RUN_WG] = tmp[q];
            __syncthreads();
        }
#pragma unroll
        for (int m = 0; m < R; m++) {
            double y = 0.0;
            if (valid[m]) {
#pragma unroll
                for (int d3 = 0; d3 < 3; d3++)
#pragma unroll
                    for (int c = 0; c < DOF; c++) y = fma(coef[m][d3 * DOF + c], xs[nbi[m][d3] + c], y);
            }
            if (lane_ok) s_part[part][m * RUN_RPB + r] = y;
        }
        __syncthreads();
        if (fin) {
            double y = s_part[0][f];
#pragma unroll
            for (int p = 1; p < 9; p++) y += s_part[p][f];
            const double dn = cheb_dn(cr.c1[s], dcur, cr.c2[s], e_di, e_b, y);
            dcur = dn;
            xo = xo + dn;
            xout[qf] = xo;  // plain store: into this XCD's L2, where every reader of the run looks
        }
        if (s + 1 == cr.nsteps) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my part of the iterate has arrived in the L2
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&ctl->cnt[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long target = (unsigned long long)(s + 1) * (unsigned long long)P;
            long spins = 0;
            int gave_up = 0;
            while (__hip_atomic_load(&ctl->cnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if ((++spins & 4095) == 0 && (spins > 2000000L || __hip_atomic_load(&ctl->gaveup[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __hip_atomic_store(&ctl->gaveup[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gave_up = 1;
                    break;
                }
            }
            s_dead = gave_up;
        }
        __syncthreads();
        if (s_dead) {
            dead = true;
            break;
        }
        const double *tmp = xout;
        xout = const_cast<double *>(xin);
        xin = tmp;
    }
    if (dead && fin) xa[qf] = xb[qf] = __builtin_nan("");
    leave();
}

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

// ---------------------------------------------------------------------------------------------------------------------
// The Lanczos run of the coarsest level (40 steps with full reorthogonalisation, the longest chain of the set-up phase:
// 287 launches, 1.3 ms of a 3.5 ms set-up at 128^3) as ONE launch on one XCD.  Same join protocol and the same matrix
// split as the Chebyshev run above; in addition every workgroup keeps ITS rows of the whole Lanczos basis in LDS, so per
// step only three things travel through the XCD's L2, each followed by one barrier:
//   (1) the partial sums of the first Gram-Schmidt pass  (P x (j+1) doubles),
//   (2) the partial sums of the second pass,
//   (3) the un-normalised new vector and the partial sums of its norm; the reader normalises and scales while staging.
// Row arithmetic as in k_dia_row_split<DOF, EPI_APPLY, 9> / k_multi_axpy / k_lanczos_next (same products, same order);
// the dot products add per-workgroup partial sums in rank order, i.e. in another order than k_multi_dot: alpha and beta
// agree with the chain of launches to rounding (tests: 1e-12 on the Ritz values), not to the bit.
constexpr int LAN_XS = 3072;     // doubles of the staged stretch (24 KB; twice: vector and D^-1/2)
constexpr int LAN_MAXS = 40;     // steps (basis of 41 vectors x R x 56 rows in LDS)
constexpr int LAN_PSTRIDE = 64;  // doubles per rank in a partial-sum region

__device__ inline int xcd_join(XcdRunCtrl *ctl, int P) {  // thread 0 of a workgroup: rank in the run or -1
    const unsigned xcc = run_xcc_id();
    const unsigned long long tk = __hip_atomic_fetch_add(&ctl->join[xcc][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tk >= (unsigned long long)P) return -1;
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
    return (w == 1ull + xcc) ? (int)tk : -1;
}
__device__ inline void xcd_leave(XcdRunCtrl *ctl) {  // all threads; the last workgroup of the launch zeroes the block
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
}
// barrier number `k` (0, 1, 2 ..) of the run: the caller's stores are drained first; returns true if the run gave up.
// (Every workgroup waits at every barrier: with ONE monotone counter a workgroup that only arrived and ran on would add
// its next arrival to the count the others are still waiting on and release them one arrival early.)
__device__ inline bool xcd_barrier(XcdRunCtrl *ctl, int k, int P, int *s_dead) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&ctl->cnt[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long target = (unsigned long long)(k + 1) * (unsigned long long)P;
        long spins = 0;
        int gave_up = 0;
        while (__hip_atomic_load(&ctl->cnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if ((++spins & 4095) == 0 && (spins > 2000000L || __hip_atomic_load(&ctl->gaveup[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(&ctl->gaveup[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gave_up = 1;
                break;
            }
        }
        *s_dead = gave_up;
    }
    __syncthreads();
    return *s_dead != 0;
}

// exch: [0, 32*64) partial sums pass 1 | [2048, 4096) pass 2 | [4096, 4096+32) norm partials | [4160, ..) the new vector
template <int DOF, int R>
__global__ __launch_bounds__(RUN_WG) void k_lanczos_run_xcd(DiaOp<DOF> op, const double *__restrict__ dinv, double *exch, double *__restrict__ alpha,
                                                            double *__restrict__ beta, int steps, XcdRunCtrl *ctl, int P) {
    constexpr int ROWS = RUN_RPB * R;
    __shared__ double s_part[9][ROWS];
    __shared__ double xs[LAN_XS], dis_s[LAN_XS];
    __shared__ double Vs[LAN_MAXS + 1][ROWS];
    __shared__ double ws[ROWS], hs[LAN_PSTRIDE];
    __shared__ double s_inv;
    __shared__ int s_dead, s_rank;
    if (threadIdx.x == 0) s_rank = xcd_join(ctl, P);
    __syncthreads();
    const int rank = s_rank;
    if (rank < 0) {
        xcd_leave(ctl);
        return;
    }
    const Geom &g = op.g;
    const long plane = g.plane();
    const long nown = g.owned_nodes() * DOF;  // == all rows (one rank; checked by the host)
    const int part = threadIdx.x / RUN_RPB, r = threadIdx.x % RUN_RPB;
    const bool lane_ok = part < 9;
    const long t_lo = (long)rank * ROWS, t_hi = min(t_lo + (long)ROWS, nown) - 1;
    const int nrows = (int)(t_hi - t_lo + 1);
    const long reach = plane + g.nx + 1;
    const long n_first = max(t_lo / DOF - reach, 0L);
    const long n_last = min(t_hi / DOF + reach, g.nodes() - 1);
    const int stage_n = (int)((n_last - n_first + 1) * DOF);
    double coef[R][3 * DOF];
    int nbi[R][3];
    bool valid[R];
#pragma unroll
    for (int m = 0; m < R; m++) {
        const long t = t_lo + (long)m * RUN_RPB + r;
        valid[m] = lane_ok && t < nown;
        const long q = valid[m] ? t : 0;
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
    for (int idx = threadIdx.x; idx < stage_n; idx += RUN_WG) dis_s[idx] = sqrt(dinv[n_first * DOF + idx]);
    const int f = threadIdx.x;
    const bool fin = f < nrows;
    const long qf = fin ? t_lo + f : 0;
    const int xsf = (int)(qf - n_first * DOF);
    double *p1 = exch, *p2 = exch + 32 * LAN_PSTRIDE, *pn = exch + 64 * LAN_PSTRIDE, *wx = exch + 64 * LAN_PSTRIDE + 64;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(wx + n_first * DOF, 0, stage_n * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_n = __builtin_amdgcn_make_buffer_rsrc(pn, 0, P * 8, 0x00020000);
    double wf = fin ? hash_u01((uint64_t)qf, 0x5eedULL) - 0.5 : 0.0;  // the start vector of k_lanczos_init
    __syncthreads();
    const double e_dis = fin ? dis_s[xsf] : 0.0;
    int nbar = 0;
    bool dead = false;
    // partial sums of this workgroup's rows against the first nv basis vectors (8 threads per vector), rank-ordered
    // total after a barrier -> hs[0 .. nv)
    auto gram = [&](int nv, double *region) -> bool {
        const int q = threadIdx.x >> 3, sub = threadIdx.x & 7;
        double s = 0.0;
        if (q < nv)
            for (int i = sub; i < nrows; i += 8) s = fma(Vs[q][i], ws[i], s);
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if (q < nv && sub == 0) region[rank * LAN_PSTRIDE + q] = s;
        if (xcd_barrier(ctl, nbar++, P, &s_dead)) return true;
        if ((int)threadIdx.x < nv) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(region, 0, P * LAN_PSTRIDE * 8, 0x00020000);
            double tmp[32];
#pragma unroll
            for (int rk = 0; rk < 32; rk++)  // ranks >= P: out of range, zeros
                tmp[rk] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (rk * LAN_PSTRIDE + threadIdx.x) * 8, 0, 16 /* sc1 */));
            double h = tmp[0];
#pragma unroll
            for (int rk = 1; rk < 32; rk++) h += tmp[rk];
            hs[threadIdx.x] = h;
        }
        __syncthreads();
        return false;
    };
    for (int j = -1; j < steps; j++) {
        // ---- (3) norm of the new vector, the vector itself
        if (fin) {
            ws[f] = wf;
            wx[qf] = wf;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            double s = 0.0;
            for (int i = threadIdx.x; i < nrows; i += 64) s = fma(ws[i], ws[i], s);
            s = wave_sum(s);
            if (threadIdx.x == 0) pn[rank] = s;
        }
        if (xcd_barrier(ctl, nbar++, P, &s_dead)) {
            dead = true;
            break;
        }
        if (threadIdx.x < 64) {
            double v = threadIdx.x < 32 ? __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs_n, threadIdx.x * 8, 0, 16)) : 0.0;
            double bb = 0.0;
            for (int rk = 0; rk < 32; rk++) bb += __shfl(v, rk);
            const double bt = sqrt(bb);
            if (threadIdx.x == 0) {
                s_inv = bt > 0.0 ? 1.0 / bt : 0.0;
                if (rank == 0 && j >= 0) beta[j] = bt;
            }
        }
        {
            constexpr int NST = LAN_XS / RUN_WG;
            double tmp[NST];
#pragma unroll
            for (int q = 0; q < NST; q++)
                tmp[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs_w, (threadIdx.x + q * RUN_WG) * 8, 0, 16 /* sc1 */));
            __syncthreads();  // s_inv
            const double inv = s_inv;
#pragma unroll
            for (int q = 0; q < NST; q++)
                if (threadIdx.x + q * RUN_WG < stage_n) xs[threadIdx.x + q * RUN_WG] = dis_s[threadIdx.x + q * RUN_WG] * (tmp[q] * inv);
            if (fin) Vs[j + 1][f] = wf * inv;
            __syncthreads();
        }
        if (j + 1 == steps) break;
        // ---- w = D^-1/2 A D^-1/2 v_{j+1}
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
            wf = e_dis * y;
            ws[f] = wf;
        }
        __syncthreads();
        // ---- (1), (2) classical Gram-Schmidt, twice, against v_0 .. v_{j+1}
        const int nv = j + 2;
        double h1_last = 0.0;
        for (int pass = 0; pass < 2 && !dead; pass++) {
            if (gram(nv, pass ? p2 : p1)) {
                dead = true;
                break;
            }
            if (fin) {
                double acc = wf;
                for (int q = 0; q < nv; q++) acc = fma(-hs[q], Vs[q][f], acc);
                wf = acc;
            }
            if (pass == 0) h1_last = hs[nv - 1];
            else if (threadIdx.x == 0 && rank == 0) alpha[nv - 1] = h1_last + hs[nv - 1];
            __syncthreads();  // hs, ws are rewritten
            if (fin) ws[f] = wf;
            __syncthreads();
        }
        if (dead) break;
    }
    if (dead && threadIdx.x == 0 && rank == 0) alpha[0] = beta[0] = __builtin_nan("");
    xcd_leave(ctl);
}

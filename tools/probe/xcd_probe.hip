// xcd_probe.hip -- what does one step of a persistent multi-workgroup Chebyshev run cost when all participants sit on
// ONE XCD (iterate exchanged through that XCD's L2: plain stores, L1-bypassing loads, arrival counter served by the same
// L2) compared with participants anywhere (round 2: agent-scope stores and counter, 2.6-3.2 us per step)?
// Join protocol: 8 * P workgroups are launched; every workgroup reads its XCC id and takes a ticket on that XCD's join
// counter; the first XCD to collect P tickets wins (pigeonhole: at least one does), its first P ticket holders take
// part, everybody else leaves.  The run does `steps` iterations of: write my rows of the iterate, barrier, read the
// whole iterate into LDS, check the values, a little arithmetic.
// build: hipcc -O3 --offload-arch=gfx950 -o xcd_probe xcd_probe.hip      run: xcd_probe [P] [steps] [rows per workgroup]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(2);                                                                   \
        }                                                                              \
    } while (0)

struct Ctrl {  // one cache line per hot word
    unsigned long long join[8][16];
    unsigned long long winner[16];   // 0: none yet, else 1 + xcc
    unsigned long long cnt[16];      // barrier arrivals
    unsigned long long bad[16];      // wrong values seen
    unsigned long long gaveup[16];
    unsigned long long xcc_of_rank[64];
};

__device__ inline unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

// MODE 0: participants anywhere (agent scope, sc1 stores / atomics at the memory side); MODE 1: one XCD (plain stores,
// arrival through a non-sc1 atomic executed in the XCD's L2, polls and data loads bypass the L1)
template <int MODE>
__global__ __launch_bounds__(512) void k_run(Ctrl *c, double *xa, double *xb, int P, int steps, int rows, long long *t_out) {
    __shared__ int s_rank;
    __shared__ double xs[8192];
    __shared__ int s_dead;
    const unsigned xcc = xcc_id();
    if (threadIdx.x == 0) {
        int rank = -1;
        if (MODE == 0) {
            const unsigned long long tk = __hip_atomic_fetch_add(&c->join[0][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            rank = tk < (unsigned long long)P ? (int)tk : -1;
        } else {
            const unsigned long long tk = __hip_atomic_fetch_add(&c->join[xcc][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tk < (unsigned long long)P) {
                if (tk == (unsigned long long)(P - 1)) {
                    unsigned long long expect = 0ull;
                    __hip_atomic_compare_exchange_strong(&c->winner[0], &expect, 1ull + xcc, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                unsigned long long w;
                long spins = 0;
                while ((w = __hip_atomic_load(&c->winner[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0ull && ++spins < 4000000L) __builtin_amdgcn_s_sleep(2);
                rank = (w == 1ull + xcc) ? (int)tk : -1;
            }
        }
        s_rank = rank;
        if (rank >= 0) c->xcc_of_rank[rank] = xcc;
    }
    __syncthreads();
    const int rank = s_rank;
    if (rank < 0) return;
    const int n = P * rows;
    const double *xin = xa;
    double *xout = xb;
    long long t0 = 0;
    unsigned long long bad = 0;
    for (int s = 0; s < steps; s++) {
        if (s == 4 && threadIdx.x == 0) t0 = wall_clock64();
        // ---- my rows of the iterate of step s: value = 1000 s + global row
        if ((int)threadIdx.x < rows) {
            const int g = rank * rows + threadIdx.x;
            const double v = 1000.0 * s + g;
            if (MODE == 0) __hip_atomic_store(&xout[g], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else xout[g] = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my stores are in the L2 (MODE 1) / on their way out (MODE 0)
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long target = (unsigned long long)(s + 1) * P;
            if (MODE == 0) __hip_atomic_fetch_add(&c->cnt[0], 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(&c->cnt[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // executed in this XCD's L2
            long spins = 0;
            int dead = 0;
            while (__hip_atomic_load(&c->cnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > 3000000L) {
                    dead = 1;
                    break;
                }
            }
            s_dead = dead;
        }
        __syncthreads();
        if (s_dead) {
            if (threadIdx.x == 0) c->gaveup[0] = 1ull + s;
            return;
        }
        // ---- everybody's rows into LDS, past the L1
        for (int i = threadIdx.x; i < n; i += 512) xs[i] = __hip_atomic_load(&xout[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 512) bad += xs[i] != 1000.0 * s + i;
        const double *t = xout;
        xout = const_cast<double *>(xin);
        xin = t;
        __syncthreads();
    }
    if (threadIdx.x == 0 && rank == 0) *t_out = wall_clock64() - t0;
    if (bad) atomicAdd((unsigned long long *)&c->bad[0], bad);
}

int main(int argc, char **argv) {
    const int P = argc > 1 ? atoi(argv[1]) : 20, steps = argc > 2 ? atoi(argv[2]) : 104, rows = argc > 3 ? atoi(argv[3]) : 112;
    Ctrl *c;
    double *xa, *xb;
    long long *t;
    CK(hipMalloc(&c, sizeof(Ctrl)));
    CK(hipMalloc(&xa, 8 * 8192));
    CK(hipMalloc(&xb, 8 * 8192));
    CK(hipMalloc(&t, 8));
    int rate = 0;
    CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));  // kHz
    for (int mode = 0; mode < 2; mode++)
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemset(c, 0, sizeof(Ctrl)));
            CK(hipMemset(xa, 0, 8 * 8192));
            CK(hipMemset(xb, 0, 8 * 8192));
            CK(hipMemset(t, 0, 8));
            if (mode == 0) hipLaunchKernelGGL(k_run<0>, dim3(P), dim3(512), 0, 0, c, xa, xb, P, steps, rows, t);
            else hipLaunchKernelGGL(k_run<1>, dim3(8 * P), dim3(512), 0, 0, c, xa, xb, P, steps, rows, t);
            CK(hipDeviceSynchronize());
            Ctrl h;
            long long ticks = 0;
            CK(hipMemcpy(&h, c, sizeof(Ctrl), hipMemcpyDeviceToHost));
            CK(hipMemcpy(&ticks, t, 8, hipMemcpyDeviceToHost));
            int xmin = 99, xmax = -1;
            for (int r = 0; r < P; r++) {
                xmin = (int)h.xcc_of_rank[r] < xmin ? (int)h.xcc_of_rank[r] : xmin;
                xmax = (int)h.xcc_of_rank[r] > xmax ? (int)h.xcc_of_rank[r] : xmax;
            }
            printf("mode %d (%s) P %d rows %d: %.3f us per step (%d steps), wrong values %llu, gave up %llu, participants on XCD %d..%d, joins per XCD:", mode,
                   mode ? "one XCD, through its L2" : "anywhere, agent scope", P, rows, ticks / (double)rate * 1e3 / (steps - 4), steps - 4,
                   (unsigned long long)h.bad[0], (unsigned long long)h.gaveup[0], xmin, xmax);
            for (int x = 0; x < 8; x++) printf(" %llu", (unsigned long long)h.join[x][0]);
            printf("\n");
        }
    return 0;
}

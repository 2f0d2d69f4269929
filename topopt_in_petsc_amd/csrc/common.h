// common.h -- shared definitions of the HIP hot path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/topopt_amd.h"

#define TP_HIP(x)                                              \
    do {                                                       \
        hipError_t e_ = (x);                                   \
        if (e_ != hipSuccess) return TP_ERR_HIP + (int)e_;     \
    } while (0)
#define TP_TRY(x)              \
    do {                       \
        int r_ = (x);          \
        if (r_) return r_;     \
    } while (0)

// Every kernel launch goes through TP_LAUNCH: a launch that the runtime rejects (LDS / register over-subscription
// after a tuning change, bad grid) must not pass silently with stale output and a plausible timing.
// TP_DEBUG_SYNC=1 additionally synchronises the device after every launch, so that an asynchronous fault is
// reported at the launch that caused it (implies no graph capture).
// The one-XCD persistent kernels (coarse_run.h: Chebyshev run, Lanczos run; coarse_direct.h: factorisation) rely on their
// workgroups being co-resident.  When one of them gives up (its peers did not show up within ~1 s: a device shared with
// another process' persistent kernels) the library redoes the work with the launch-per-step forms and keeps the
// one-XCD forms OFF for the rest of the process (topopt_amd.hip: redo_without_xcd).
inline bool &tp_xcd_disabled() {
    static bool off = false;
    return off;
}
// Round 6: the coarse factorisation, a chain of one-XCD kernels, runs on a side stream BESIDE the head of the
// solve (mg.h: cd_pending) -- full-device kernels are queued while its later kernels start, so their co-residency is a
// little less certain than on an idle device.  A give-up of THAT chain alone first costs the deferral (the factorisation is
// joined at the end of the set-up again, round 4's behaviour), not the one-XCD forms; only a give-up without the deferral
// switches them off.  Every recovery is counted (tp_xcd_status).
inline bool &tp_defer_disabled() {
    static bool off = false;
    return off;
}
inline int &tp_giveup_count() {
    static int n = 0;
    return n;
}
// test switch TP_TEST_FORCE_GIVEUP = "<mode>" or "<mode>:<rank>": does it ask rank `rank` for recovery branch `mode`?
// rank < 0: does it ask ANY rank (the collective agreement must then be reached by all of them)
inline bool tp_test_force_giveup(int mode, int rank) {
    const char *e = getenv("TP_TEST_FORCE_GIVEUP");
    if (!e || atoi(e) != mode) return false;
    const char *c = strchr(e, ':');
    return rank < 0 || !c || atoi(c + 1) == rank;
}
inline bool tp_debug_sync() {
    static const bool v = getenv("TP_DEBUG_SYNC") != nullptr && atoi(getenv("TP_DEBUG_SYNC")) != 0;
    return v;
}
inline int tp_launch_check(const char *what) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && tp_debug_sync()) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        fprintf(stderr, "topopt_amd: launch failed (%s): %s\n", what, hipGetErrorString(e));
        return TP_ERR_HIP + (int)e;
    }
    return TP_OK;
}
#define TP_LAUNCH(kernel, ...)                            \
    do {                                                  \
        hipLaunchKernelGGL(kernel, __VA_ARGS__);          \
        int lrc_ = tp_launch_check(#kernel);              \
        if (lrc_) return lrc_;                            \
    } while (0)

constexpr int WAVE = 64;     // CDNA wavefront
constexpr int BLK  = 256;    // default workgroup: 4 waves, one per SIMD
constexpr int MAX_RED_BLOCKS = 8192;

// One multigrid level of the local z-slab.  Local node plane k <-> global plane
// gz0 + k.  Element layer l spans node planes l, l+1.
struct Geom {
    int nx, ny, nzl;      // local node planes stored (own + ghosts)
    int ex, ey;           // elements per row / column
    int ez_own;           // own element layers [0, ez_own)
    int ezl;              // element layers with data available locally (own + ghost above)
    int own_lo, own_hi;   // owned node planes, inclusive
    int gz0;              // global z index of local node plane 0
    int nz_glob;          // global node planes on this level
    int has_lo, has_hi;   // neighbour slabs
    __host__ __device__ long plane() const { return (long)nx * ny; }
    __host__ __device__ long nodes() const { return (long)nx * ny * nzl; }
    __host__ __device__ long owned_nodes() const { return (long)nx * ny * (own_hi - own_lo + 1); }
    __host__ __device__ long own_elems() const { return (long)ex * ey * ez_own; }
    __host__ __device__ long elems_stored() const { return (long)ex * ey * ezl; }
};

// local corner offsets of the 8-node hexahedron, reference node order
// (LinearElasticity.cc:819-826): counter-clockwise in the lower plane, then the upper.
__device__ __constant__ int c_LX[8] = {0, 1, 1, 0, 0, 1, 1, 0};
__device__ __constant__ int c_LY[8] = {0, 0, 1, 1, 0, 0, 1, 1};
__device__ __constant__ int c_LZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};
static const int h_LX[8] = {0, 1, 1, 0, 0, 1, 1, 0};
static const int h_LY[8] = {0, 0, 1, 1, 0, 0, 1, 1};
static const int h_LZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};
// compile-time versions for fully unrolled loops
__host__ __device__ constexpr int LXc(int a) { return (a == 1 || a == 2 || a == 5 || a == 6) ? 1 : 0; }
__host__ __device__ constexpr int LYc(int a) { return (a == 2 || a == 3 || a == 6 || a == 7) ? 1 : 0; }
__host__ __device__ constexpr int LZc(int a) { return a >= 4 ? 1 : 0; }
// corner index from (lx,ly,lz)
__host__ __device__ constexpr int corner_of(int lx, int ly, int lz) {
    return lz * 4 + (ly ? (lx ? 2 : 3) : (lx ? 1 : 0));
}

// ---------------------------------------------------------------------------
// deterministic reductions: wave shuffle -> LDS -> one partial per workgroup,
// then a single-workgroup pass over the partials.  No atomics: run-to-run
// bit-reproducible.
// ---------------------------------------------------------------------------
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    return v;
}

// returns the block total in thread 0 (BLK threads)
__device__ inline double block_sum(double v) {
    __shared__ double s_part[BLK / WAVE];
    v = wave_sum(v);
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    __syncthreads();  // protect s_part reuse across consecutive calls
    if (lane == 0) s_part[w] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < BLK / WAVE; i++) t += s_part[i];
    }
    return t;
}

// Tail of a reducing kernel: every workgroup has its NV partial sums in thread 0 (block_sum); they are written to
// partials[v * nb + b] and the LAST workgroup to arrive adds them up in the same fixed order as k_reduce_final --
// bitwise the same result, one launch less per dot product.  Hand-off per MI355X_MICROARCH.md ("8-byte agent atomics
// both sides"): the partial goes out as a relaxed agent-scope atomic store (write-through, sc1), is drained with an
// explicit vmcnt(0), then the relaxed agent atomic ticket; the last arriver reads the partials with relaxed
// agent-scope atomic loads (served past its L1).  NO release fence: `buffer_wbl2` would write back the XCD's whole
// dirty L2 once per workgroup -- measured +40 % on the design iteration when every block of the CG update did it.
// Arrivals are counted on 8 shard counters (workgroup b -> shard b & 7, i.e. its XCD under round-robin dispatch) and
// the last arriver of a shard on a top counter: same-address atomics serialise at ~12 ns each, which a single counter
// turns into 100 us for an 8192-workgroup grid -- and so do counters that share a cache line (one atomic unit serves
// the line: 8 adjacent words measured like one, 28 us for a 2048-workgroup dot product), hence TICKET_STRIDE.
// The counters are left at 0.
constexpr int TICKET_STRIDE = 1024;              // unsigned words between two counters (4 KB: another channel)
constexpr int TICKET_WORDS = 9 * TICKET_STRIDE;  // allocation: 8 shards + top
template <int NV>
__device__ inline void reduce_tail(const double (&mine)[NV], double *__restrict__ partials, int nb, int b,
                                   unsigned *ticket, double *__restrict__ out, double *__restrict__ out_host = nullptr) {
    __shared__ int s_last;
    if (!ticket) {  // TP_NO_REDUCE_TAIL=1: partial sums only, the host launches k_reduce_final behind this kernel
        if (threadIdx.x == 0) {
#pragma unroll
            for (int v = 0; v < NV; v++) partials[(long)v * nb + b] = mine[v];
        }
        return;
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int v = 0; v < NV; v++)
            __hip_atomic_store(&partials[(long)v * nb + b], mine[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int sh = b & 7;
        const unsigned in_shard = (unsigned)((nb + 7 - sh) >> 3);  // workgroups b' < nb with b' & 7 == sh
        int last = 0;
        unsigned *mine_t = ticket + sh * TICKET_STRIDE, *top_t = ticket + 8 * TICKET_STRIDE;
        if (__hip_atomic_fetch_add(mine_t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_shard - 1) {
            __hip_atomic_store(mine_t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned shards = (unsigned)(nb < 8 ? nb : 8);
            last = __hip_atomic_fetch_add(top_t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == shards - 1;
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
#pragma unroll
    for (int v = 0; v < NV; v++) {
        double s = 0.0;
        // eight loads in flight per thread, added in index order (the order of the plain loop): the loads go past
        // the L2 to memory, one at a time they cost the last workgroup ~1.5 us each
        for (int i0 = threadIdx.x; i0 < nb; i0 += 8 * BLK) {
            double w[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = i0 + k * BLK;
                w[k] = __hip_atomic_load(&partials[(long)v * nb + (i < nb ? i : i0)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (i0 + k * BLK < nb) s += w[k];
        }
        s = block_sum(s);
        if (threadIdx.x == 0) {
            out[v] = s;
            if (out_host) out_host[v] = s;  // pinned host memory: the host reads it behind an event, no copy in the stream
        }
    }
    if (threadIdx.x == 0) __hip_atomic_store(ticket + 8 * TICKET_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// partials laid out [value][block]; out[v] = sum_b partials[v*nblocks + b]
template <int NV>
__global__ __launch_bounds__(BLK) void k_reduce_final(const double *__restrict__ partials, int nblocks,
                                                      double *__restrict__ out) {
    for (int v = 0; v < NV; v++) {
        double s = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += BLK) s += partials[(long)v * nblocks + b];
        s = block_sum(s);
        if (threadIdx.x == 0) out[v] = s;
    }
}

// ---------------------------------------------------------------------------
// BLAS-1 style kernels (grid-stride, coefficients read from device scalars so
// that the Krylov loop needs no host round trip for them)
// ---------------------------------------------------------------------------
inline int grid_for(long n, int cap = 2048) {
    long b = (n + BLK - 1) / BLK;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}

__global__ __launch_bounds__(BLK) void k_set(double *__restrict__ x, double a, long n) {
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) x[i] = a;
}
__global__ __launch_bounds__(BLK) void k_scale(double *__restrict__ x, double a, long n) {
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) x[i] *= a;
}
__global__ __launch_bounds__(BLK) void k_copy(double *__restrict__ y, const double *__restrict__ x, long n) {
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) y[i] = x[i];
}
// y = a*x*z (pointwise), used by Lanczos scaling
__global__ __launch_bounds__(BLK) void k_pw_mult(double *__restrict__ y, const double *__restrict__ x,
                                                 const double *__restrict__ z, long n) {
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) y[i] = x[i] * z[i];
}
// partial dot products: partials[0*nb+b] = sum a*b
// y = a x + b y ;  w = x (.*|./) y
__global__ __launch_bounds__(BLK) void k_axpby(double *__restrict__ y, double a, const double *__restrict__ x, double b, long n) {
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) y[i] = a * x[i] + b * y[i];
}
__global__ __launch_bounds__(BLK) void k_pw_div(double *__restrict__ w, const double *__restrict__ x,
                                                const double *__restrict__ y, long n) {
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) w[i] = x[i] / y[i];
}
__global__ __launch_bounds__(BLK) void k_dot(const double *__restrict__ a, const double *__restrict__ b, long n,
                                             double *__restrict__ partials, unsigned *ticket, double *__restrict__ out) {
    double s = 0.0;
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) s += a[i] * b[i];
    const double v[1] = {block_sum(s)};
    reduce_tail<1>(v, partials, gridDim.x, blockIdx.x, ticket, out);
}
__global__ __launch_bounds__(BLK) void k_sum(const double *__restrict__ a, long n, double *__restrict__ partials,
                                             unsigned *ticket, double *__restrict__ out) {
    double s = 0.0;
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) s += a[i];
    const double v[1] = {block_sum(s)};
    reduce_tail<1>(v, partials, gridDim.x, blockIdx.x, ticket, out);
}

// splitmix64 -> [0,1): the synthetic-field generator of SURVEY.md 8(d)
__host__ __device__ inline double hash_u01(uint64_t idx, uint64_t seed) {
    uint64_t z = (idx + 1u) * 0x9E3779B97F4A7C15ULL + seed;
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// step_probe.hip -- which part of a coarse-run step (coarse_run.h) costs what: staging loads past L2, LDS work, the
// write-through stores, the release arrival, the polling.  build: hipcc -O3 --offload-arch=gfx950 -o step_probe step_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int WG = 512, NST = 12;
template <int STAGE, int STORE, int REL, int SC>
__global__ __launch_bounds__(WG) void k_steps(unsigned long long *cnt, double *xa, double *xb, int rounds, int stage_n, double *sink) {
    __shared__ double xs[WG * NST];
    const double *xin = xa; double *xout = xb;
    double acc = 0.0;
    const long base_off = (long)blockIdx.x * 224;
    for (int s = 0; s < rounds; s++) {
        if (STAGE) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(xin) + base_off, 0, stage_n * 8, 0x00020000);
            double tmp[NST];
#pragma unroll
            for (int q = 0; q < NST; q++) tmp[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (threadIdx.x + q * WG) * 8, 0, SC));
#pragma unroll
            for (int q = 0; q < NST; q++) if (threadIdx.x + q * WG < stage_n) xs[threadIdx.x + q * WG] = tmp[q];
        }
        __syncthreads();
        double y = 0.0;
#pragma unroll
        for (int q = 0; q < 36; q++) y = fma(1.0 + q, xs[(threadIdx.x * 3 + q * 7) % (stage_n > 0 ? stage_n : 1)], y);
        acc += y;
        __syncthreads();
        if (STORE && threadIdx.x < 224) {
            if (STORE == 1) __hip_atomic_store(&xout[base_off + 300 + threadIdx.x], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else xout[base_off + 300 + threadIdx.x] = acc;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (REL == 1) __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else if (REL == 0) { __builtin_amdgcn_s_waitcnt(0); __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            if (REL != 2) {
                const unsigned long long target = (unsigned long long)(s + 1) * gridDim.x;
                long spins = 0;
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) if (++spins > 2000000L) break;
            }
        }
        __syncthreads();
        const double *t = xout; xout = const_cast<double *>(xin); xin = t;
    }
    if (acc == 12345.678) sink[0] = acc;
}
template <int STAGE, int STORE, int REL, int SC>
void run(const char *name, unsigned long long *cnt, double *xa, double *xb, double *sink) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int rounds = 1000;
    for (int n : {1, 10, 32}) {
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipMemset(cnt, 0, 16);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL((k_steps<STAGE, STORE, REL, SC>), dim3(n), dim3(WG), 0, 0, cnt, xa, xb, rounds, 771, sink);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-64s %2d workgroups: %.2f us per step\n", name, n, best * 1e3 / rounds);
    }
}
int main() {
    unsigned long long *cnt; double *xa, *xb, *sink;
    (void)hipMalloc(&cnt, 16); (void)hipMalloc(&xa, 8 << 20); (void)hipMalloc(&xb, 8 << 20); (void)hipMalloc(&sink, 8);
    (void)hipMemset(xa, 0, 8 << 20); (void)hipMemset(xb, 0, 8 << 20);
    run<0, 0, 2, 16>("LDS work + 4 syncthreads only (no barrier)", cnt, xa, xb, sink);
    run<1, 0, 2, 16>("+ staging loads sc1 (no barrier)", cnt, xa, xb, sink);
    run<1, 0, 2, 0>("+ staging loads through L2 (no barrier)", cnt, xa, xb, sink);
    run<1, 1, 2, 16>("+ staging sc1 + agent stores (no barrier)", cnt, xa, xb, sink);
    run<0, 0, 0, 16>("barrier only (waitcnt + relaxed add)", cnt, xa, xb, sink);
    run<0, 0, 1, 16>("barrier only (release add)", cnt, xa, xb, sink);
    run<0, 1, 1, 16>("agent stores + release barrier", cnt, xa, xb, sink);
    run<0, 2, 1, 16>("plain stores + release barrier", cnt, xa, xb, sink);
    run<1, 1, 1, 16>("the whole step: staging sc1 + agent stores + release barrier", cnt, xa, xb, sink);
    run<1, 1, 0, 16>("the whole step with waitcnt + relaxed add", cnt, xa, xb, sink);
    run<1, 2, 1, 16>("the whole step with plain stores", cnt, xa, xb, sink);
    return 0;
}

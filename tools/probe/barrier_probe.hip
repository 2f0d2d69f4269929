// barrier_probe.hip -- what does an in-kernel barrier between N resident workgroups cost on MI355X, against the ~6.5 us
// of a dependent kernel launch?  Every round: each workgroup publishes 64 doubles (agent-scope stores), arrives at a
// counter (release), waits for all, then reads its neighbour's 64 doubles (agent-scope loads) and checks them.
// build: hipcc -O3 --offload-arch=gfx950 -o barrier_probe barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_rounds(unsigned *cnt, double *buf, int rounds, unsigned long long *errs, int fence_mode) {
    const int w = blockIdx.x, n = gridDim.x;
    unsigned long long bad = 0;
    for (int r = 0; r < rounds; r++) {
        if (threadIdx.x < 64) __hip_atomic_store(&buf[(size_t)(r & 1) * n * 64 + w * 64 + threadIdx.x], r * 1000.0 + w + threadIdx.x * 1e-3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x == 0) {
            if (fence_mode) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else {
                __builtin_amdgcn_s_waitcnt(0);  // own stores performed (they are write-through, agent scope)
                __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const unsigned target = (unsigned)(r + 1) * n;
            long spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                // a workgroup that never arrives (not co-resident): give up everywhere instead of hanging the device
                if (++spins > 2000000L || __hip_atomic_load(errs + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(errs + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __syncthreads();
        if (__hip_atomic_load(errs + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            if (threadIdx.x == 0) atomicAdd(errs, 1000000ull);
            return;
        }
        if (threadIdx.x < 64) {
            const int nbw = (w + 1) % n;
            const double v = __hip_atomic_load(&buf[(size_t)(r & 1) * n * 64 + nbw * 64 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != r * 1000.0 + nbw + threadIdx.x * 1e-3) bad++;
        }
    }
    if (bad) atomicAdd(errs, bad);
}
__global__ void k_tiny(double *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0; }

int main() {
    unsigned *cnt; double *buf; unsigned long long *errs;
    hipMalloc(&cnt, 4); hipMalloc(&buf, sizeof(double) * 2 * 1024 * 64); hipMalloc(&errs, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int rounds = 2000;
    for (int mode = 0; mode < 2; mode++)
        for (int n : {1, 8, 16, 32, 64, 128, 256, 512}) {
            for (int threads : {256, 1024}) {
                if (n * threads > 256 * 2048) continue;
                float best = 1e9;
                unsigned long long herr = 0;
                for (int rep = 0; rep < 3; rep++) {
                    hipMemset(cnt, 0, 4); hipMemset(errs, 0, 16);
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(k_rounds, dim3(n), dim3(threads), 0, 0, cnt, buf, rounds, errs, mode);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                    unsigned long long h; hipMemcpy(&h, errs, 8, hipMemcpyDeviceToHost); herr += h;
                }
                printf("mode %s  %3d workgroups x %4d threads: %.2f us per round, errors %llu\n", mode ? "release-add" : "waitcnt+relaxed-add", n, threads, best * 1e3 / rounds, herr);
            }
        }
    // the alternative: a chain of dependent tiny launches
    hipEventRecord(e0);
    for (int i = 0; i < 2000; i++) hipLaunchKernelGGL(k_tiny, dim3(64), dim3(256), 0, 0, buf);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("dependent tiny launches: %.2f us each\n", ms * 1e3 / 2000);
    return 0;
}

// fine_probe.hip -- stand-alone check and timing of the fine-level operator kernels (no Python, no torch):
// k_fine_dma<EPI, TX, TY, D> (csrc/fine_dma.h) against k_fine_tile<EPI> (csrc/fine_tile.h), bit for bit, on a
// synthetic cantilever-like problem, then HIP-event timings of both at the given size.
// build: tools/probe/build_fine_probe.sh      run: fine_probe ex ey ez [reps] [kz] [mode]
//   mode bit 0: bit checks, bit 1: timings, bit 2: all tile shapes / depths (default: 16x16, D = 2 only)
#include "../../topopt_in_petsc_amd/csrc/elements.h"
#include "../../topopt_in_petsc_amd/csrc/fine_u4.h"

#include <chrono>
#include <string>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(2);                                                                   \
        }                                                                              \
    } while (0)

struct Prob {
    int ex, ey, ez, nx, ny, nz;
    long nn, ne;
    double *u, *b, *E, *y0, *y1, *init;
    uint8_t *mask, *colmask;
    double *partials, *red;
    unsigned *ticket;
    int slot;
};

static void fill(std::vector<double> &v, uint64_t seed, double lo, double hi) {
    for (size_t i = 0; i < v.size(); i++) v[i] = lo + (hi - lo) * hash_u01(i, seed);
}

static Prob make(int ex, int ey, int ez, bool with_mask) {
    Prob p{};
    p.ex = ex, p.ey = ey, p.ez = ez, p.nx = ex + 1, p.ny = ey + 1, p.nz = ez + 1;
    p.nn = (long)p.nx * p.ny * p.nz, p.ne = (long)ex * ey * ez;
    std::vector<double> h(3 * p.nn);
    CK(hipMalloc(&p.u, 24 * p.nn));
    CK(hipMalloc(&p.b, 24 * p.nn));
    CK(hipMalloc(&p.y0, 24 * p.nn));
    CK(hipMalloc(&p.y1, 24 * p.nn));
    CK(hipMalloc(&p.init, 24 * p.nn));
    CK(hipMalloc(&p.E, 8 * p.ne));
    fill(h, 1, -1.0, 1.0);
    CK(hipMemcpy(p.u, h.data(), 24 * p.nn, hipMemcpyHostToDevice));
    fill(h, 2, -1.0, 1.0);
    CK(hipMemcpy(p.b, h.data(), 24 * p.nn, hipMemcpyHostToDevice));
    fill(h, 3, -1.0, 1.0);
    CK(hipMemcpy(p.init, h.data(), 24 * p.nn, hipMemcpyHostToDevice));
    std::vector<double> e(p.ne);
    for (long i = 0; i < p.ne; i++) {
        double x = 0.05 + 0.95 * hash_u01(i, 7);
        if (getenv("PROBE_SIMP")) {  // the bench's synthetic density: smooth field + noise, clamped at 1e-3
            const double xc = ((i % ex) + 0.5) / ey, yc = (((i / ex) % ey) + 0.5) / ey, zc = ((i / ((long)ex * ey)) + 0.5) / ey;
            x = 0.12 + 0.4 * sin(7 * M_PI * xc) * sin(5 * M_PI * yc) * sin(3 * M_PI * zc) + 0.3 * (hash_u01(i, 12345) - 0.5);
            x = x < 1e-3 ? 1e-3 : (x > 1.0 ? 1.0 : x);
        }
        e[i] = 1e-9 + x * x * x * (1.0 - 1e-9);
    }
    CK(hipMemcpy(p.E, e.data(), 8 * p.ne, hipMemcpyHostToDevice));
    p.mask = p.colmask = nullptr;
    if (with_mask) {  // cantilever: face x = 0 clamped; plus a few scattered single dofs
        std::vector<uint8_t> m(p.nn, 0), cm((long)p.nx * p.ny, 0);
        for (long n = 0; n < p.nn; n++) {
            const int i = (int)(n % p.nx);
            if (i == 0) m[n] = 7;
            else if (!getenv("PROBE_FACE_ONLY") && hash_u01(n, 11) < 2e-4) m[n] = (uint8_t)(1 + (int)(hash_u01(n, 12) * 6.99));
            cm[n % ((long)p.nx * p.ny)] |= m[n];
        }
        CK(hipMalloc(&p.mask, p.nn));
        CK(hipMalloc(&p.colmask, cm.size()));
        CK(hipMemcpy(p.mask, m.data(), p.nn, hipMemcpyHostToDevice));
        CK(hipMemcpy(p.colmask, cm.data(), cm.size(), hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&p.partials, 8 * 65536));
    CK(hipMalloc(&p.red, 64));
    CK(hipMalloc(&p.ticket, 4 * TICKET_WORDS));
    CK(hipMemset(p.ticket, 0, 4 * TICKET_WORDS));
    double KE[576];
    const double hh = 1.0 / ey;
    hex8_stiffness_box(hh, hh, hh, 0.3, KE);
    SymKE sk;
    const double dropped = make_sym_ke(KE, &sk);
    if (dropped > 1e-12) fprintf(stderr, "KE not box symmetric: %g\n", dropped);
    p.slot = sym_slot_acquire(sk);
    return p;
}

static int balanced(const Prob &p, int kz) {  // equal chunks: ceil(planes / number of chunks)
    const int nch = (p.nz + kz - 1) / kz;
    return (p.nz + nch - 1) / nch;
}
static TileArgs targs(const Prob &p, int kz) {
    return TileArgs{p.nx, p.ny, p.nz, p.ex, p.ey, p.ez, 0, p.nz - 1, kz, p.E, p.mask, p.colmask, p.slot * SYMKE_STRIDE, 0, 0, nullptr, 1, 0, 0, -1,
                    0, nullptr, nullptr, 0, nullptr};
}
static NodeArgs nargs(const Prob &p, double *out, int epi, bool prev) {
    NodeArgs a{};
    a.x = p.u, a.out = out, a.b = p.b, a.d = nullptr, a.dinv = nullptr;
    a.c1 = prev ? 0.37 : 0.0, a.c2 = 0.81, a.prev_zero = 0;
    a.partials = p.partials, a.ticket = p.ticket, a.red_out = p.red;
    (void)epi;
    return a;
}

template <int EPI>
static void launch_old(const Prob &p, int kz, double *out, bool prev) {
    const int tx = (p.nx + TOUT - 1) / TOUT, ty = (p.ny + TOUT - 1) / TOUT, tz = (p.nz + kz - 1) / kz;
    hipLaunchKernelGGL((k_fine_tile<EPI>), dim3(tx, ty, tz), dim3(TILE * TILE), 0, 0, targs(p, kz), nargs(p, out, EPI, prev));
}
template <int EPI, int TX, int TY, int D, int WPS, bool CARRY = false>
static void launch_new(const Prob &p, int kz, double *out, bool prev) {
    using S = FineDma<TX, TY, D>;
    kz = balanced(p, kz);
    const int tx = (p.nx + S::TOX - 1) / S::TOX, ty = (p.ny + S::TOY - 1) / S::TOY, tz = (p.nz + kz - 1) / kz;
    hipLaunchKernelGGL((k_fine_dma<EPI, TX, TY, D, WPS, CARRY>), dim3(tx, ty, tz), dim3(TX * TY), 0, 0, targs(p, kz), nargs(p, out, EPI, prev));
}

template <int EPI, int TX, int TY, int WPS, bool CARRY>
static void launch_u4(const Prob &p, int kz, double *out, bool prev) {
    using S = FineU4<TX, TY>;
    kz = balanced(p, kz);
    const int tx = (p.nx + S::TOX - 1) / S::TOX, ty = (p.ny + S::TOY - 1) / S::TOY, tz = (p.nz + kz - 1) / kz;
    hipLaunchKernelGGL((k_fine_u4<EPI, TX, TY, WPS, CARRY>), dim3(tx, ty, tz), dim3(TX * TY), 0, 0, targs(p, kz), nargs(p, out, EPI, prev));
}

static double bytes_of(const Prob &p, int epi, bool prev) {
    const bool cheb = epi == EPI_CHEB || epi == EPI_CHEB_DOT;
    const double vecs = epi == EPI_APPLY || epi == EPI_APPLY_DOT ? 2 : (epi == EPI_RESID ? 3 : (prev ? 4 : 3));
    (void)cheb;
    return 24.0 * vecs * p.nn + 8.0 * p.ne;
}

template <class F>
static double time_us(F f, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3 * ms / reps;
}

static long diff_count(const Prob &p, double *red_ref, bool dot) {
    std::vector<double> a(3 * p.nn), b(3 * p.nn);
    CK(hipMemcpy(a.data(), p.y0, 24 * p.nn, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), p.y1, 24 * p.nn, hipMemcpyDeviceToHost));
    long bad = 0, first = -1;
    for (long i = 0; i < 3 * p.nn; i++)
        if (memcmp(&a[i], &b[i], 8) != 0) {
            if (first < 0) first = i;
            bad++;
        }
    if (bad && getenv("PROBE_DIAG")) {  // where do the differences sit?
        std::vector<double> hb(3 * p.nn), hu(3 * p.nn);
        CK(hipMemcpy(hb.data(), p.b, 24 * p.nn, hipMemcpyDeviceToHost));
        long by_c[3] = {0, 0, 0}, by_tx[16] = {0}, by_ty[16] = {0}, by_z[64] = {0};
        int shown = 0;
        for (long i = 0; i < 3 * p.nn; i++)
            if (memcmp(&a[i], &b[i], 8) != 0) {
                const long n = i / 3;
                const int ii = (int)(n % p.nx), jj = (int)((n / p.nx) % p.ny), kk = (int)(n / ((long)p.nx * p.ny));
                by_c[i % 3]++, by_tx[ii % 15]++, by_ty[jj % 15]++, by_z[kk % 64]++;
                if (shown++ < 12)
                    fprintf(stderr, "     dof %ld node (%d,%d,%d) c%ld: old %.17g new %.17g diff %.3g   b %.17g  b-old %.17g b-new %.17g\n", i, ii, jj, kk, i % 3, a[i], b[i],
                            b[i] - a[i], hb[i], hb[i] - a[i], hb[i] - b[i]);
            }
        fprintf(stderr, "     by component: %ld %ld %ld\n     by x %% 15:", by_c[0], by_c[1], by_c[2]);
        for (int q = 0; q < 15; q++) fprintf(stderr, " %ld", by_tx[q]);
        fprintf(stderr, "\n     by y %% 15:");
        for (int q = 0; q < 15; q++) fprintf(stderr, " %ld", by_ty[q]);
        fprintf(stderr, "\n     by z:");
        for (int q = 0; q < 40; q++) fprintf(stderr, " %ld", by_z[q]);
        fprintf(stderr, "\n");
    }
    if (bad) {
        const long n = first / 3;
        fprintf(stderr, "   first difference at dof %ld (node %ld,%ld,%ld c%ld): %.17g vs %.17g\n", first, n % p.nx, (n / p.nx) % p.ny, n / ((long)p.nx * p.ny),
                first % 3, a[first], b[first]);
    }
    if (dot) {
        double r;
        CK(hipMemcpy(&r, p.red, 8, hipMemcpyDeviceToHost));
        if (fabs(r - *red_ref) > 1e-12 * fabs(*red_ref)) {
            fprintf(stderr, "   reduction differs: %.17g vs %.17g\n", *red_ref, r);
            bad++;
        }
    }
    return bad;
}

template <int EPI, int TX, int TY, int D, int WPS, bool CARRY = false>
static int check(const Prob &p, int kz_old, int kz_new, bool prev, const char *name) {
    constexpr bool dot = EPI == EPI_APPLY_DOT || EPI == EPI_CHEB_DOT;
    CK(hipMemcpy(p.y0, p.init, 24 * p.nn, hipMemcpyDeviceToDevice));
    CK(hipMemcpy(p.y1, p.init, 24 * p.nn, hipMemcpyDeviceToDevice));
    launch_old<EPI>(p, kz_old, p.y0, prev);
    CK(hipDeviceSynchronize());
    double red_ref = 0;
    if (dot) CK(hipMemcpy(&red_ref, p.red, 8, hipMemcpyDeviceToHost));
    launch_new<EPI, TX, TY, D, WPS, CARRY>(p, kz_new, p.y1, prev);
    {
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            printf("check %-28s LAUNCH FAILED: %s\n", name, hipGetErrorString(e));
            return 1;
        }
    }
    CK(hipDeviceSynchronize());
    const long bad = diff_count(p, &red_ref, dot);
    printf("check %-28s %dx%dx%d mask=%d kz %d/%d prev=%d: %s (%ld)\n", name, p.ex, p.ey, p.ez, p.mask != nullptr, kz_old, kz_new, (int)prev,
           bad ? "DIFFERENT" : "bit-identical", bad);
    fflush(stdout);
    return bad != 0;
}

template <int EPI, int TX, int TY, int D, int WPS, bool CARRY = false>
static void timing(const Prob &p, int kz, bool prev, int reps, const char *name) {
    using S = FineDma<TX, TY, D>;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (k_fine_dma<EPI, TX, TY, D, WPS, CARRY>), TX * TY, 0) != hipSuccess) occ = -1;
    launch_new<EPI, TX, TY, D, WPS, CARRY>(p, kz, p.y1, prev);
    {
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            printf("time %-30s LAUNCH FAILED: %s\n", name, hipGetErrorString(e));
            return;
        }
    }
    const double us = time_us([&] { launch_new<EPI, TX, TY, D, WPS, CARRY>(p, kz, p.y1, prev); }, reps);
    const double gb = bytes_of(p, EPI, prev) / 1e9;
    printf("time %-30s kz %3d lds %6d wg/CU %d : %8.1f us  %7.1f GB/s  frac %.3f\n", name, kz, S::LDS_BYTES, occ, us, gb / us * 1e6, gb / us * 1e6 / 8000.0);
    fflush(stdout);
}
template <int EPI, int TX, int TY, int WPS, bool CARRY>
static int check_u4(const Prob &p, int kz_old, int kz_new, bool prev, const char *name) {
    constexpr bool dot = EPI == EPI_APPLY_DOT || EPI == EPI_CHEB_DOT;
    CK(hipMemcpy(p.y0, p.init, 24 * p.nn, hipMemcpyDeviceToDevice));
    CK(hipMemcpy(p.y1, p.init, 24 * p.nn, hipMemcpyDeviceToDevice));
    launch_old<EPI>(p, kz_old, p.y0, prev);
    CK(hipDeviceSynchronize());
    double red_ref = 0;
    if (dot) CK(hipMemcpy(&red_ref, p.red, 8, hipMemcpyDeviceToHost));
    launch_u4<EPI, TX, TY, WPS, CARRY>(p, kz_new, p.y1, prev);
    {
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            printf("check %-28s LAUNCH FAILED: %s\n", name, hipGetErrorString(e));
            return 1;
        }
    }
    CK(hipDeviceSynchronize());
    const long bad = diff_count(p, &red_ref, dot);
    printf("check %-28s %dx%dx%d mask=%d kz %d/%d prev=%d: %s (%ld)\n", name, p.ex, p.ey, p.ez, p.mask != nullptr, kz_old, kz_new, (int)prev,
           bad ? "DIFFERENT" : "bit-identical", bad);
    fflush(stdout);
    return bad != 0;
}
template <int EPI, int TX, int TY, int WPS, bool CARRY>
static void timing_u4(const Prob &p, int kz, bool prev, int reps, const char *name) {
    using S = FineU4<TX, TY>;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (k_fine_u4<EPI, TX, TY, WPS, CARRY>), TX * TY, 0) != hipSuccess) occ = -1;
    launch_u4<EPI, TX, TY, WPS, CARRY>(p, kz, p.y1, prev);
    {
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            printf("time %-30s LAUNCH FAILED: %s\n", name, hipGetErrorString(e));
            return;
        }
    }
    const double us = time_us([&] { launch_u4<EPI, TX, TY, WPS, CARRY>(p, kz, p.y1, prev); }, reps);
    const double gb = bytes_of(p, EPI, prev) / 1e9;
    printf("time %-30s kz %3d lds %6d wg/CU %d : %8.1f us  %7.1f GB/s  frac %.3f\n", name, balanced(p, kz), S::LDS_BYTES, occ, us, gb / us * 1e6, gb / us * 1e6 / 8000.0);
    fflush(stdout);
}
template <int EPI>
static void timing_old(const Prob &p, int kz, bool prev, int reps, const char *name) {
    const double us = time_us([&] { launch_old<EPI>(p, kz, p.y1, prev); }, reps);
    const double gb = bytes_of(p, EPI, prev) / 1e9;
    printf("time %-30s kz %3d                    : %8.1f us  %7.1f GB/s  frac %.3f\n", name, kz, us, gb / us * 1e6, gb / us * 1e6 / 8000.0);
    fflush(stdout);
}

// what the box streams for the SpMV's byte mix (read u and E, write y; 16 B per lane, grid-stride): the ceiling of any tiling
__global__ __launch_bounds__(256) void k_stream_ref(const double2 *__restrict__ u, const double2 *__restrict__ E, double2 *__restrict__ y, long nu, long ne) {
    const long stride = (long)gridDim.x * 256;
    double2 acc = make_double2(0.0, 0.0);
    for (long i = blockIdx.x * 256l + threadIdx.x; i < ne; i += stride) {
        const double2 e = E[i];
        acc.x += e.x, acc.y += e.y;
    }
    for (long i = blockIdx.x * 256l + threadIdx.x; i < nu; i += stride) {
        double2 v = u[i];
        v.x += acc.x, v.y += acc.y;
        y[i] = v;
    }
}


int main(int argc, char **argv) {
    const int ex = argc > 1 ? atoi(argv[1]) : 64, ey = argc > 2 ? atoi(argv[2]) : 64, ez = argc > 3 ? atoi(argv[3]) : 64;
    const int reps = argc > 4 ? atoi(argv[4]) : 10;
    const int kz = argc > 5 ? atoi(argv[5]) : 16;
    const int mode = argc > 6 ? atoi(argv[6]) : 3;
    const bool ext = (mode & 4) != 0;  // bit 2: the other tile shapes and depths too
    const bool u4 = (mode & 8) != 0;   // bit 3: the unrolled kernel (fine_u4.h)
    int fails = 0;
    if (mode & 1) {
        for (int with_mask = 0; with_mask < 2; with_mask++) {
            Prob p = make(ex, ey, ez, with_mask != 0);
            fails += check<EPI_APPLY, 16, 16, 2, 3>(p, 8, kz, false, "apply 16x16 D2");
            fails += check<EPI_RESID, 16, 16, 2, 3>(p, 8, kz, false, "resid 16x16 D2");
            fails += check<EPI_CHEB, 16, 16, 2, 3>(p, 8, kz, true, "cheb 16x16 D2");
            fails += check<EPI_CHEB, 16, 16, 2, 3>(p, 8, kz, false, "cheb(c1=0) 16x16 D2");
            fails += check<EPI_APPLY_DOT, 16, 16, 2, 3>(p, 8, kz, false, "apply_dot 16x16 D2");
            fails += check<EPI_CHEB_DOT, 16, 16, 2, 3>(p, 8, kz, true, "cheb_dot 16x16 D2");
#define U4_VARIANTS(F)                                                   \
    F(EPI_APPLY, 16, 16, 3, true, false, "u4 apply 16x16 carry")         \
    F(EPI_CHEB, 16, 16, 2, true, true, "u4 cheb 16x16 w2 carry")         \
    F(EPI_APPLY, 32, 8, 2, true, false, "u4 apply 32x8 carry w2")        \
    F(EPI_CHEB, 32, 8, 2, true, true, "u4 cheb 32x8 w2 carry")           \
    F(EPI_APPLY, 64, 4, 2, true, false, "u4 apply 64x4 carry w2")        \
    F(EPI_CHEB, 64, 4, 2, true, true, "u4 cheb 64x4 w2 carry")
#define CHKU(E, TX, TY, W, C, PV, NAME) fails += check_u4<E, TX, TY, W, C>(p, 8, kz, PV, NAME);
            if (u4) { U4_VARIANTS(CHKU) }
            if (ext) {
#define EXT_VARIANTS(F)                                                   \
    F(EPI_APPLY, 16, 16, 1, 4, false, false, "apply 16x16 D1 w4")          \
    F(EPI_APPLY, 16, 16, 2, 3, true, false, "apply 16x16 D2 carry")       \
    F(EPI_CHEB, 16, 16, 2, 3, true, true, "cheb 16x16 D2 carry")          \
    F(EPI_APPLY, 16, 16, 3, 2, false, false, "apply 16x16 D3")            \
    F(EPI_APPLY, 32, 8, 1, 4, false, false, "apply 32x8 D1 w4")           \
    F(EPI_APPLY, 32, 16, 1, 4, false, false, "apply 32x16 D1 w4")         \
    F(EPI_APPLY, 32, 16, 1, 3, true, false, "apply 32x16 D1 carry")       \
    F(EPI_APPLY, 32, 24, 2, 3, true, false, "apply 32x24 D2 carry")       \
    F(EPI_CHEB, 32, 24, 2, 3, false, true, "cheb 32x24 D2")               \
    F(EPI_APPLY, 32, 32, 1, 4, false, false, "apply 32x32 D1 w4")         \
    F(EPI_CHEB, 32, 32, 1, 4, false, true, "cheb 32x32 D1 w4")            \
    F(EPI_CHEB, 32, 16, 1, 3, false, true, "cheb 32x16 D1 (w3)")           \
    F(EPI_APPLY, 32, 24, 2, 3, false, false, "apply 32x24 D2")            \
    F(EPI_CHEB, 32, 8, 1, 3, false, true, "cheb 32x8 D1")
#define CHK(E, TX, TY, D, W, C, PV, NAME) fails += check<E, TX, TY, D, W, C>(p, 8, kz, PV, NAME);
                EXT_VARIANTS(CHK)
            }
        }
    }
    if (mode & 2) {
        Prob p = make(ex, ey, ez, getenv("PROBE_FACE_ONLY") != nullptr);
        const int kzs[3] = {kz, 2 * kz, 4 * kz};
        const int nkz = getenv("PROBE_ONE_KZ") ? 1 : 3;
        for (int nb = 2048; nb <= 32768; nb *= 4) {
            const double us = time_us([&] { hipLaunchKernelGGL(k_stream_ref, dim3(nb), dim3(256), 0, 0, (const double2 *)p.u, (const double2 *)p.E, (double2 *)p.y1, 3 * p.nn / 2, p.ne / 2); }, reps);
            printf("time stream reference (%5d wgs)                        : %8.1f us  %7.1f GB/s  frac %.3f\n", nb, us, bytes_of(p, EPI_APPLY, false) / us * 1e-3, bytes_of(p, EPI_APPLY, false) / us * 1e-3 / 8000.0);
        }
        timing_old<EPI_APPLY>(p, 16, false, reps, "old apply");
        timing_old<EPI_CHEB>(p, 16, true, reps, "old cheb");
        for (int q = 0; q < nkz; q++) {
            const int k = kzs[q];
            timing<EPI_APPLY, 16, 16, 2, 3>(p, k, false, reps, "apply 16x16 D2");
            timing<EPI_CHEB, 16, 16, 2, 3>(p, k, true, reps, "cheb 16x16 D2");
#define TIMU(E, TX, TY, W, C, PV, NAME) timing_u4<E, TX, TY, W, C>(p, k, PV, reps, NAME);
            if (u4) { U4_VARIANTS(TIMU) }
            if (ext) {
#define TIM(E, TX, TY, D, W, C, PV, NAME) timing<E, TX, TY, D, W, C>(p, k, PV, reps, NAME);
                EXT_VARIANTS(TIM)
            }
        }
    }
    printf("%s\n", fails ? "FAILURES" : "all checks passed");
    return fails ? 1 : 0;
}

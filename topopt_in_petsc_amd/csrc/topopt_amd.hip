// topopt_amd.hip -- C-ABI entry points (include/topopt_amd.h) of the MI355X-native
// hot path.  gfx950 only; no CPU fallback anywhere in this library.
#include <string>

#include "elements.h"
#include "mg.h"
#include "refksp.h"
#include "rccl_comm.h"

// ===========================================================================
// grid
// ===========================================================================
extern "C" int tp_grid_create(tp_grid **out, const tp_grid_opts *o) {
    if (!out || !o) return TP_ERR_ARG;
    if (o->nx < 2 || o->ny < 2 || o->nz < 2 || o->nranks < 1 || o->rank < 0 || o->rank >= o->nranks) return TP_ERR_ARG;
    if ((o->nz - 1) % o->nranks) return TP_ERR_ARG;
    if (o->nranks > 1 && !o->comm) return TP_ERR_ARG;
    TP_HIP(hipSetDevice(o->device));
    tp_grid *g = new tp_grid();
    g->o = *o;
    g->stream = (hipStream_t)o->stream;
    g->has_comm = o->nranks > 1;
    if (g->has_comm) {
        g->comm = g->comm_host = *o->comm;
        // second stream for the overlapped halos (TP_OVERLAP=0: every halo on the grid's stream, before its consumer)
        const bool want = !(getenv("TP_OVERLAP") && atoi(getenv("TP_OVERLAP")) == 0);
        if (want && hipStreamCreateWithFlags(&g->comm_stream, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&g->ev_ready, hipEventDisableTiming) == hipSuccess)
            g->overlap = true;
        else
            (void)hipGetLastError();
    }
    g->ex = o->nx - 1;
    g->ey = o->ny - 1;
    g->ez_glob = o->nz - 1;
    g->ez_own = g->ez_glob / o->nranks;
    g->rank = o->rank;
    g->nranks = o->nranks;
    g->alg_bytes = g->flops = 0.0;
    g->launches = 0;
    Geom q = make_geom(g, 0);
    long nblk = q.nodes() / BLK + 2;
    if (nblk < MAX_RED_BLOCKS) nblk = MAX_RED_BLOCKS;
    if (hipMalloc((void **)&g->partials, sizeof(double) * 4 * (size_t)nblk) != hipSuccess ||
        hipMalloc((void **)&g->scal, sizeof(double) * 64) != hipSuccess ||
        hipMalloc((void **)&g->ticket, sizeof(unsigned) * TICKET_WORDS) != hipSuccess ||
        hipHostMalloc((void **)&g->h_scal, sizeof(double) * 64) != hipSuccess) {
        delete g;
        return TP_ERR_HIP + (int)hipErrorOutOfMemory;
    }
    g->h_scal_dev = nullptr;
    if (hipHostGetDevicePointer((void **)&g->h_scal_dev, g->h_scal, 0) != hipSuccess) {
        (void)hipGetLastError();
        g->h_scal_dev = nullptr;  // (the solver then copies its scalars back as before)
    }
    (void)hipMemsetAsync(g->scal, 0, sizeof(double) * 64, g->stream);
    (void)hipMemsetAsync(g->ticket, 0, sizeof(unsigned) * TICKET_WORDS, g->stream);
    double W[512];
    host_W(W);
    TP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_W), W, sizeof(W)));
    *out = g;
    return TP_OK;
}
extern "C" int tp_rccl_load(const char *path) { return rccl_load(path); }
extern "C" int tp_rccl_unique_id(void *id128) {
    if (!rccl_api().handle || !id128) return TP_ERR_STATE;
    ncclUniqueId id;
    if (rccl_api().GetUniqueId(&id) != ncclSuccess) return TP_ERR_COMM;
    memcpy(id128, &id, sizeof(id));
    return TP_OK;
}
extern "C" int tp_grid_use_rccl2(tp_grid *g, const void *id128, const void *id128_halo) {
    if (!g || !id128 || !g->has_comm || g->rccl) return TP_ERR_ARG;
    if (!rccl_api().handle) return TP_ERR_STATE;
    RcclComm *c = nullptr;
    const int rc = rccl_comm_create(&c, id128, g->rank, g->nranks, g->o.device, g->stream, g->comm.cap, id128_halo);
    if (rc) return rc;
    g->rccl = c;
    g->comm = c->hooks;
    return TP_OK;
}
extern "C" int tp_grid_use_rccl(tp_grid *g, const void *id128) { return tp_grid_use_rccl2(g, id128, nullptr); }
extern "C" int tp_grid_drop_rccl(tp_grid *g) {
    if (!g) return TP_ERR_ARG;
    if (g->rccl) {
        g->comm = g->comm_host;
        rccl_comm_destroy(g->rccl);
        g->rccl = nullptr;
    }
    return TP_OK;
}
// Rank-tagged planes through the CURRENT hooks (staged and in-place exchange, all-reduce, all-gather) and a check
// of what arrives: *ok = 1 if this rank saw exactly its neighbours' data.  Collective over the grid's ranks.
__global__ __launch_bounds__(BLK) void k_selftest_fill(double *a, double *b, long n, uint64_t seed) {
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) {
        a[i] = hash_u01((uint64_t)i, seed) - 0.5;
        b[i] = hash_u01((uint64_t)i, seed + 77) + 0.25;
    }
}
extern "C" int tp_grid_reduction_selftest(tp_grid *g, long n, int reps, int *mismatches) {
    if (!g || n < 1 || reps < 1 || reps > 4096 || !mismatches) return TP_ERR_ARG;
    double *a = nullptr, *b = nullptr, *res = nullptr;
    struct Guard {  // the early returns of TP_HIP / TP_LAUNCH free the buffers too
        double *&a, *&b, *&r;
        hipStream_t s;
        ~Guard() {
            (void)hipStreamSynchronize(s);
            (void)hipFree(a);
            (void)hipFree(b);
            (void)hipFree(r);
        }
    } guard{a, b, res, g->stream};
    TP_HIP(hipMalloc((void **)&a, sizeof(double) * (size_t)n));
    TP_HIP(hipMalloc((void **)&b, sizeof(double) * (size_t)n));
    TP_HIP(hipMalloc((void **)&res, sizeof(double) * 2 * (size_t)reps));
    const int nb = grid_for(n, 2048);
    for (int r = 0; r < reps; r++) {
        if (r % 16 == 0) TP_LAUNCH(k_selftest_fill, dim3(grid_for(n)), dim3(BLK), 0, g->stream, a, b, n, (uint64_t)(1000 + r));
        // the in-kernel tail (or whatever TP_NO_REDUCE_TAIL leaves of it) ...
        TP_LAUNCH(k_dot, dim3(nb), dim3(BLK), 0, g->stream, a, b, n, g->partials, tail_ticket(g), res + 2 * r);
        if (!tail_ticket(g)) TP_LAUNCH(k_reduce_final<1>, dim3(1), dim3(BLK), 0, g->stream, g->partials, nb, res + 2 * r);
        // ... and the two-launch form
        TP_LAUNCH(k_dot, dim3(nb), dim3(BLK), 0, g->stream, a, b, n, g->partials, (unsigned *)nullptr, res + 2 * r + 1);
        TP_LAUNCH(k_reduce_final<1>, dim3(1), dim3(BLK), 0, g->stream, g->partials, nb, res + 2 * r + 1);
    }
    std::vector<double> h(2 * (size_t)reps);
    TP_HIP(hipMemcpyAsync(h.data(), res, sizeof(double) * h.size(), hipMemcpyDeviceToHost, g->stream));
    TP_HIP(hipStreamSynchronize(g->stream));
    int bad = 0;
    for (int r = 0; r < reps; r++) bad += std::memcmp(&h[2 * r], &h[2 * r + 1], sizeof(double)) != 0 || !(h[2 * r] == h[2 * r]);
    *mismatches = bad;
    return TP_OK;
}
extern "C" int tp_grid_comm_selfcheck(tp_grid *g, int *ok) {
    if (!g || !ok) return TP_ERR_ARG;
    *ok = 1;
    if (!g->has_comm) return TP_OK;
    const tp_comm &c = g->comm;
    hipStream_t st = g->stream;
    const long n = c.cap < 4096 ? c.cap : 4096;
    const int nr = g->nranks, rk = g->rank;
    std::vector<double> h(n), a(n), b(n);
    auto fill = [&](double *dst, double tag) {
        for (long i = 0; i < n; i++) h[i] = tag + 1e-3 * (double)i;
        return hipMemcpyAsync(dst, h.data(), sizeof(double) * n, hipMemcpyHostToDevice, st) == hipSuccess &&
               hipStreamSynchronize(st) == hipSuccess;
    };
    auto check = [&](const double *src, double tag) {
        if (hipMemcpyAsync(a.data(), src, sizeof(double) * n, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
        if (hipStreamSynchronize(st) != hipSuccess) return false;
        for (long i = 0; i < n; i++)
            if (a[i] != tag + 1e-3 * (double)i) return false;
        return true;
    };
    // A hook that fails on THIS rank must not end the sequence: the other ranks are inside the same collectives and
    // would block.  Failures are recorded, every step runs on every rank, the verdict is returned at the end (the
    // host then takes the minimum over ranks).
    bool hook_failed = false;
    bool good = fill(c.send_lo, 100.0 * rk + 1.0) && fill(c.send_hi, 100.0 * rk + 2.0);
    if (c.exchange(c.user, n)) hook_failed = true;
    if (rk > 0) good = good && check(c.recv_lo, 100.0 * (rk - 1) + 2.0);       // lower neighbour's send_hi
    if (rk < nr - 1) good = good && check(c.recv_hi, 100.0 * (rk + 1) + 1.0);  // upper neighbour's send_lo
    if (c.exchange_direct) {  // in place: use the staging areas as "vectors", crossed over
        good = good && fill(c.recv_lo, 100.0 * rk + 3.0) && fill(c.recv_hi, 100.0 * rk + 4.0);
        const int rc = c.exchange_direct(c.user, rk > 0 ? c.recv_lo : nullptr, rk > 0 ? c.send_lo : nullptr,
                                         rk < nr - 1 ? c.recv_hi : nullptr, rk < nr - 1 ? c.send_hi : nullptr, n);
        if (rc == 0) {
            if (rk > 0) good = good && check(c.send_lo, 100.0 * (rk - 1) + 4.0);
            if (rk < nr - 1) good = good && check(c.send_hi, 100.0 * (rk + 1) + 3.0);
        } else if (rc == 2) {
            g->comm.exchange_direct = nullptr;  // the host cannot address our memory in place: staged exchange from now on
        } else {
            hook_failed = true;
        }
    }
    {   // sum over ranks of (rank + 1) in slot 0..3
        for (int i = 0; i < 4; i++) h[i] = (double)(rk + 1) * (i + 1);
        good = good && hipMemcpyAsync(c.red, h.data(), sizeof(double) * 4, hipMemcpyHostToDevice, st) == hipSuccess;
        if (c.allreduce_sum(c.user, 4)) hook_failed = true;
        good = good && hipMemcpyAsync(b.data(), c.red, sizeof(double) * 4, hipMemcpyDeviceToHost, st) == hipSuccess &&
               hipStreamSynchronize(st) == hipSuccess;
        for (int i = 0; i < 4; i++) good = good && b[i] == 0.5 * nr * (nr + 1) * (i + 1);
        if (c.allreduce_inplace) {
            good = good && hipMemcpyAsync(c.red + 8, h.data(), sizeof(double) * 4, hipMemcpyHostToDevice, st) == hipSuccess;
            if (c.allreduce_inplace(c.user, c.red + 8, 4)) hook_failed = true;
            good = good && hipMemcpyAsync(b.data(), c.red + 8, sizeof(double) * 4, hipMemcpyDeviceToHost, st) == hipSuccess &&
                   hipStreamSynchronize(st) == hipSuccess;
            for (int i = 0; i < 4; i++) good = good && b[i] == 0.5 * nr * (nr + 1) * (i + 1);
        }
    }
    if (c.allgather) {
        good = good && fill(c.send_lo, 1000.0 * rk);
        if (c.allgather(c.user, n)) hook_failed = true;
        for (int r = 0; r < nr; r++) good = good && check(c.gather + (long)r * n, 1000.0 * r);
    }
    *ok = (good && !hook_failed) ? 1 : 0;
    return hook_failed ? TP_ERR_COMM : TP_OK;
}
extern "C" long tp_grid_overlapped_halos(const tp_grid *g) { return g ? g->n_overlapped : 0; }
// HIP-event timing of the roofline kernel where it runs: on = 1 starts collecting a pair of events around every launch
// of the fine level's fused Chebyshev step (one rank, unsplit launches); the read synchronises the stream, returns the
// sum of the pairs' elapsed times and their number, and clears them
extern "C" int tp_grid_kernel_timer(tp_grid *g, int on) {
    if (!g) return TP_ERR_ARG;
    g->kt_on = on != 0;
    if (on) g->kt_bytes = 0.0;
    return TP_OK;
}
extern "C" int tp_grid_kernel_timer_read(tp_grid *g, double *total_ms, long *launches);
// ... and the algorithmic bytes of exactly those launches (the first step of a smoothing sweep reads one vector less)
extern "C" int tp_grid_kernel_timer_read2(tp_grid *g, double *total_ms, long *launches, double *alg_bytes) {
    if (!g || !alg_bytes) return TP_ERR_ARG;
    const double b = g->kt_bytes;
    const int rc = tp_grid_kernel_timer_read(g, total_ms, launches);
    if (rc == TP_OK) *alg_bytes = b;
    return rc;
}
extern "C" int tp_grid_kernel_timer_read(tp_grid *g, double *total_ms, long *launches) {
    if (!g || !total_ms || !launches) return TP_ERR_ARG;
    TP_HIP(hipStreamSynchronize(g->stream));
    double t = 0.0;
    long n = 0;
    for (size_t i = 0; i + 1 < g->kt_ev.size(); i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g->kt_ev[i], g->kt_ev[i + 1]) == hipSuccess) {
            t += ms;
            n++;
        }
    }
    for (hipEvent_t e : g->kt_ev) (void)hipEventDestroy(e);
    g->kt_ev.clear();
    g->kt_bytes = 0.0;
    *total_ms = t;
    *launches = n;
    return TP_OK;
}
// Timing of the communication where it runs (N > 1; grid.h: CommMark).  on = 1 starts collecting; the read synchronises both
// streams and returns, per kind (0 blocking halo, 1 overlapped halo, 2 all-reduce, 3 all-gather), the number of hook calls, the
// host wall time spent inside them and the device time between the event pairs around them, then clears the collection.
extern "C" int tp_grid_comm_timer(tp_grid *g, int on) {
    if (!g) return TP_ERR_ARG;
    g->ct_on = on != 0;
    return TP_OK;
}
extern "C" int tp_grid_comm_timer_read(tp_grid *g, long calls[4], double host_ms[4], double device_ms[4]) {
    if (!g || !calls || !host_ms || !device_ms) return TP_ERR_ARG;
    TP_HIP(hipStreamSynchronize(g->stream));
    if (g->comm_stream) TP_HIP(hipStreamSynchronize(g->comm_stream));
    for (int k = 0; k < 4; k++) {
        double t = 0.0;
        for (size_t i = 0; i + 1 < g->ct_ev[k].size(); i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, g->ct_ev[k][i], g->ct_ev[k][i + 1]) == hipSuccess) t += ms;
        }
        for (hipEvent_t e : g->ct_ev[k]) (void)hipEventDestroy(e);
        g->ct_ev[k].clear();
        calls[k] = g->ct_calls[k];
        host_ms[k] = 1e3 * g->ct_host_s[k];
        device_ms[k] = t;
        g->ct_calls[k] = 0;
        g->ct_host_s[k] = 0.0;
    }
    return TP_OK;
}
extern "C" int tp_grid_comm_stats(const tp_grid *g, long *ex, long *red) {
    if (!g) return TP_ERR_ARG;
    if (ex) *ex = g->rccl ? g->rccl->n_exchanges : 0;
    if (red) *red = g->rccl ? g->rccl->n_reductions : 0;
    return TP_OK;
}
// what the in-library RCCL path is made of, for the bench line: ranks the COMMUNICATOR reports (ncclCommCount; -1 if the
// entry point is missing, 0 without RCCL), and whether the neighbour exchanges have a communicator of their own
extern "C" int tp_grid_comm_info(const tp_grid *g, int *rccl_ranks, int *two_communicators) {
    if (!g) return TP_ERR_ARG;
    int n = 0;
    if (g->rccl && g->rccl->comm) {
        n = -1;
        if (rccl_api().CommCount && rccl_api().CommCount(g->rccl->comm, &n) != ncclSuccess) n = -1;
    }
    if (rccl_ranks) *rccl_ranks = n;
    if (two_communicators) *two_communicators = g->rccl && g->rccl->comm_halo ? 1 : 0;
    return TP_OK;
}
__global__ void k_selftest_fill(double *p, long n, double base) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = base + (double)i;
}
extern "C" int tp_rccl_selftest(int device, void *stream, long n, double *max_err) {
    if (!rccl_api().handle || n < 1 || !max_err) return TP_ERR_STATE;
    ncclUniqueId id, id2;
    if (rccl_api().GetUniqueId(&id) != ncclSuccess || rccl_api().GetUniqueId(&id2) != ncclSuccess) return TP_ERR_COMM;
    RcclComm *c = nullptr;
    hipStream_t st = (hipStream_t)stream;
    int rc = rccl_comm_create(&c, &id, 0, 1, device, st, n < 16 ? 16 : n, &id2);  // exchanges on their own communicator
    if (rc) return rc;
    c->periodic = true;
    tp_comm &h = c->hooks;
    std::vector<double> a(n), b(n);
    double err = 0.0;
    // staged exchange: send_hi arrives in recv_lo, send_lo in recv_hi
    TP_LAUNCH(k_selftest_fill, dim3(64), dim3(256), 0, st, h.send_lo, n, 1000.0);
    TP_LAUNCH(k_selftest_fill, dim3(64), dim3(256), 0, st, h.send_hi, n, 5000.0);
    rc = h.exchange(h.user, n);
    if (!rc) {
        (void)hipMemcpyAsync(a.data(), h.recv_lo, sizeof(double) * n, hipMemcpyDeviceToHost, st);
        (void)hipMemcpyAsync(b.data(), h.recv_hi, sizeof(double) * n, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        for (long i = 0; i < n; i++) err = fmax(err, fmax(fabs(a[i] - (5000.0 + i)), fabs(b[i] - (1000.0 + i))));
        // in-place variant on other buffers (gather area as scratch: 1 rank -> cap doubles)
        TP_LAUNCH(k_selftest_fill, dim3(64), dim3(256), 0, st, h.recv_lo, n, 7000.0);
        TP_LAUNCH(k_selftest_fill, dim3(64), dim3(256), 0, st, h.recv_hi, n, 9000.0);
        rc = h.exchange_direct(h.user, h.recv_lo, h.send_lo, h.recv_hi, h.send_hi, n);  // to_lo, from_lo, to_hi, from_hi
    }
    if (!rc) {
        (void)hipMemcpyAsync(a.data(), h.send_lo, sizeof(double) * n, hipMemcpyDeviceToHost, st);  // from_lo <- to_hi
        (void)hipMemcpyAsync(b.data(), h.send_hi, sizeof(double) * n, hipMemcpyDeviceToHost, st);  // from_hi <- to_lo
        (void)hipStreamSynchronize(st);
        for (long i = 0; i < n; i++) err = fmax(err, fmax(fabs(a[i] - (9000.0 + i)), fabs(b[i] - (7000.0 + i))));
        TP_LAUNCH(k_selftest_fill, dim3(1), dim3(64), 0, st, h.red, 16, 3.0);
        rc = h.allreduce_sum(h.user, 8);
        if (!rc) rc = h.allreduce_inplace(h.user, h.red + 8, 8);
    }
    if (!rc) {
        TP_LAUNCH(k_selftest_fill, dim3(64), dim3(256), 0, st, h.send_lo, n, 11000.0);
        rc = h.allgather(h.user, n);
    }
    if (!rc) {
        double r16[16];
        (void)hipMemcpyAsync(r16, h.red, sizeof(r16), hipMemcpyDeviceToHost, st);
        (void)hipMemcpyAsync(a.data(), h.gather, sizeof(double) * n, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        for (int i = 0; i < 16; i++) err = fmax(err, fabs(r16[i] - (3.0 + i)));
        for (long i = 0; i < n; i++) err = fmax(err, fabs(a[i] - (11000.0 + i)));
    }
    // The overlapped-halo pattern (grid.h: halo_nodes_begin): in-place exchanges issued on a SECOND stream, ordered
    // by events against fills on the first one, interleaved with all-reduces on the first stream -- several rounds,
    // the way a smoother chain issues them.
    if (!rc) {
        hipStream_t cs = nullptr;
        hipEvent_t ready = nullptr, done = nullptr;
        if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess)
            rc = TP_ERR_HIP;
        for (int round = 0; round < 4 && !rc; round++) {
            TP_LAUNCH(k_selftest_fill, dim3(64), dim3(256), 0, st, h.recv_lo, n, 100.0 * round + 1.0);  // "to_lo" plane
            TP_LAUNCH(k_selftest_fill, dim3(64), dim3(256), 0, st, h.recv_hi, n, 100.0 * round + 2.0);  // "to_hi" plane
            (void)hipEventRecord(ready, st);
            (void)hipStreamWaitEvent(cs, ready, 0);
            h.set_stream(h.user, cs);
            rc = h.exchange_direct(h.user, h.recv_lo, h.send_lo, h.recv_hi, h.send_hi, n);  // from_lo <- to_hi, from_hi <- to_lo
            h.set_stream(h.user, nullptr);
            (void)hipEventRecord(done, cs);
            TP_LAUNCH(k_selftest_fill, dim3(1), dim3(64), 0, st, h.red, 8, 5.0);   // "interior" work + a reduction meanwhile
            if (!rc) rc = h.allreduce_inplace(h.user, h.red, 8);
            (void)hipStreamWaitEvent(st, done, 0);
            (void)hipMemcpyAsync(a.data(), h.send_lo, sizeof(double) * n, hipMemcpyDeviceToHost, st);
            (void)hipMemcpyAsync(b.data(), h.send_hi, sizeof(double) * n, hipMemcpyDeviceToHost, st);
            if (hipStreamSynchronize(st) != hipSuccess) rc = TP_ERR_HIP;
            for (long i = 0; i < n && !rc; i++)
                err = fmax(err, fmax(fabs(a[i] - (100.0 * round + 2.0 + i)), fabs(b[i] - (100.0 * round + 1.0 + i))));
        }
        if (cs) {
            (void)hipStreamSynchronize(cs);
            (void)hipStreamDestroy(cs);
        }
        if (ready) (void)hipEventDestroy(ready);
        if (done) (void)hipEventDestroy(done);
    }
    rccl_comm_destroy(c);
    *max_err = err;
    return rc ? TP_ERR_COMM : TP_OK;
}
extern "C" int tp_grid_destroy(tp_grid *g) {
    if (!g) return TP_OK;
    (void)hipStreamSynchronize(g->stream);
    rccl_comm_destroy(g->rccl);
    if (g->comm_stream) {
        (void)hipStreamSynchronize(g->comm_stream);
        (void)hipStreamDestroy(g->comm_stream);
    }
    if (g->ev_ready) (void)hipEventDestroy(g->ev_ready);
    if (g->ev_scal) (void)hipEventDestroy(g->ev_scal);
    for (hipEvent_t e : g->kt_ev) (void)hipEventDestroy(e);  // a kernel timer that was never read
    (void)hipFree(g->partials);
    (void)hipFree(g->scal);
    (void)hipFree(g->ticket);
    (void)hipHostFree(g->h_scal);
    delete g;
    return TP_OK;
}
extern "C" long tp_grid_local_nodes(const tp_grid *g) { return make_geom(g, 0).nodes(); }
extern "C" long tp_grid_local_elems(const tp_grid *g) { return make_geom(g, 0).own_elems(); }
extern "C" long tp_grid_owned_node_offset(const tp_grid *g) {
    Geom q = make_geom(g, 0);
    return q.plane() * q.own_lo;
}
extern "C" long tp_grid_owned_nodes(const tp_grid *g) { return make_geom(g, 0).owned_nodes(); }
extern "C" int tp_grid_node_z0(const tp_grid *g) { return make_geom(g, 0).gz0; }
extern "C" int tp_grid_elem_z0(const tp_grid *g) { return g->rank * g->ez_own; }
extern "C" int tp_grid_halo_nodes(tp_grid *g, double *v, int dof) {
    if (!g || !v || dof < 1) return TP_ERR_ARG;
    return halo_nodes(g, make_geom(g, 0), v, dof);
}

extern "C" int tp_set_device(int device) {
    TP_HIP(hipSetDevice(device));
    return TP_OK;
}
extern "C" int tp_malloc(void **p, size_t bytes) {
    TP_HIP(hipMalloc(p, bytes));
    return TP_OK;
}
extern "C" int tp_free(void *p) {
    TP_HIP(hipFree(p));
    return TP_OK;
}
extern "C" int tp_memcpy_h2d(void *dst, const void *src, size_t bytes) {
    TP_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return TP_OK;
}
extern "C" int tp_memcpy_d2h(void *dst, const void *src, size_t bytes) {
    TP_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return TP_OK;
}
extern "C" int tp_sync(const tp_grid *g) {
    TP_HIP(hipStreamSynchronize(g->stream));
    return TP_OK;
}
extern "C" int tp_vec_scale(tp_grid *g, double *x, double a, long n) {
    TP_LAUNCH(k_scale, dim3(grid_for(n)), dim3(BLK), 0, g->stream, x, a, n);
    count_launch(g, 16.0 * n, 1.0 * n);
    return TP_OK;
}
// BLAS-1 surface of the PETSc-named adapter (include/petsc_shim.h): Vec operations of LinearElasticity.cc / Filter.cc
extern "C" int tp_vec_axpby(tp_grid *g, double *y, double a, const double *x, double b, long n) {
    TP_LAUNCH(k_axpby, dim3(grid_for(n)), dim3(BLK), 0, g->stream, y, a, x, b, n);
    count_launch(g, 24.0 * n, 3.0 * n);
    return TP_OK;
}
extern "C" int tp_vec_pointwise(tp_grid *g, double *w, const double *x, const double *y, int divide, long n) {
    if (divide)
        TP_LAUNCH(k_pw_div, dim3(grid_for(n)), dim3(BLK), 0, g->stream, w, x, y, n);
    else
        TP_LAUNCH(k_pw_mult, dim3(grid_for(n)), dim3(BLK), 0, g->stream, w, x, y, n);
    count_launch(g, 24.0 * n, 1.0 * n);
    return TP_OK;
}
// sum over ranks of x.y (y = NULL: of x) over n local entries -- the caller passes OWNED ranges only
extern "C" int tp_vec_dot(tp_grid *g, const double *x, const double *y, long n, double *out) {
    if (!g || !x || !out) return TP_ERR_ARG;
    TP_TRY(y ? dot_to_slot(g, x, y, n, S_TMP) : sum_to_slot(g, x, n, S_TMP));
    return read_scal(g, S_TMP, 1, out);
}
extern "C" int tp_vec_set(tp_grid *g, double *x, double a, long n) {
    TP_LAUNCH(k_set, dim3(grid_for(n)), dim3(BLK), 0, g->stream, x, a, n);
    count_launch(g, 8.0 * n, 0.0);
    return TP_OK;
}

__global__ __launch_bounds__(BLK) void k_synth_density(int ex, int ey, int ez, int e0z, double h, uint64_t seed,
                                                       double *__restrict__ x) {
    const long n = (long)ex * ey * ez;
    const double pi = 3.14159265358979323846;
    for (long t = blockIdx.x * (long)BLK + threadIdx.x; t < n; t += (long)gridDim.x * BLK) {
        const int i = (int)(t % ex), j = (int)((t / ex) % ey), k = (int)(t / ((long)ex * ey));
        const uint64_t gid = (uint64_t)i + (uint64_t)ex * ((uint64_t)j + (uint64_t)ey * (uint64_t)(k + e0z));
        const double xc = (i + 0.5) * h, yc = (j + 0.5) * h, zc = (k + e0z + 0.5) * h;
        double v = 0.12 + 0.4 * sin(7 * pi * xc) * sin(5 * pi * yc) * sin(3 * pi * zc) + 0.3 * (hash_u01(gid, seed) - 0.5);
        v = v < 1e-3 ? 1e-3 : (v > 1.0 ? 1.0 : v);
        x[t] = v;
    }
}
extern "C" int tp_synth_density(tp_grid *g, double *x, uint64_t seed) {
    const long n = (long)g->ex * g->ey * g->ez_own;
    TP_LAUNCH(k_synth_density, dim3(grid_for(n)), dim3(BLK), 0, g->stream, g->ex, g->ey, g->ez_own,
                       g->rank * g->ez_own, g->o.hy, seed, x);
    count_launch(g);
    return TP_OK;
}

// ===========================================================================
// linear elasticity
// ===========================================================================
extern "C" int tp_abi_version(void) { return TP_ABI_VERSION; }
extern "C" unsigned long tp_solver_opts_size(void) { return (unsigned long)sizeof(tp_solver_opts); }
extern "C" void tp_solver_default_opts(tp_solver_opts *o) {
    o->nlvls = 4;       // LinearElasticity.cc:23
    o->nu = 0.3;        // :22
    o->rtol = 1.0e-5;   // :621
    o->atol = 1.0e-50;  // :622
    o->dtol = 1.0e5;    // :623
    o->max_it = 200;    // :625
    o->nsmooth = 4;     // :635
    o->ncoarse = 30;    // :631
    o->cheb_lo = 0.1;   // PETSc's default Chebyshev window 0.1 / 1.1 of the estimate
    o->cheb_hi = 1.1;
    o->nlanczos = 10;
    o->fine_eig = 0;
    o->ksp_mode = 0;
    o->restart = 100;        // :624
    o->smooth_pc = 1;        // :745 PCSOR
    o->coarse_pc = 1;        // :731 PCSOR
    o->coarse_restart = 30;  // :632
    o->coarse_rtol = 1.0e-8; // :628
    o->coarse_direct = 0;
}

struct tp_elasticity {
    tp_grid *grid;
    MGSolver<3> mg;
    hipStream_t aux_stream = nullptr;            // level 2's Galerkin kernel beside level 1's (assemble)
    hipEvent_t aux_fork = nullptr, aux_done = nullptr;
    double KE[576];
    double *d_KE, *d_M;      // element matrix, 8 child matrices (level 0 -> 1 fast path)
    double *d_E;             // SIMP moduli, own + ghost-above layer
    uint8_t *d_mask;         // clamped-dof bits per local node
    uint8_t *d_colmask;      // OR of d_mask over the planes of a node column
    std::vector<uint8_t> h_mask;
    int *d_flagged;          // level-1 elements touching clamped nodes
    int *d_flag_all;         // flagged level-1 elements incl. the ghost layer (matrix-free level 1)
    int nflag_all;
    int *d_corr_nodes, *d_corr_adj;
    double *d_dK, *d_corr, *d_corr_tmp;
    // matrix-free level 1 keeps no element matrices but those of the flagged elements (compact rows: own flagged in
    // list order, then the ghost-layer ones received from the upper neighbour) and the element -> row map
    double *d_KelF;
    int *d_fidx1;
    double *d_M2;            // 64 grand-child matrices (level 0 -> 2 in one step)
    uint8_t *d_flag2;        // own level-2 elements with a flagged level-1 child
    int *d_list2;            // ... as a list
    int nlist2;
    int nx_first;            // max over ranks of the flagged elements in a rank's first own level-1 layer
    int nflagged;
    double *d_bN;            // RHS .* N scratch
    double *d_N;             // copy of N (for the load masking of :542)
    bool have_bc, assembled;
};

__global__ __launch_bounds__(BLK) void k_simp(const double *__restrict__ x, double Emin, double Emax, double penal,
                                              double *__restrict__ E, long n) {
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK)
        E[i] = Emin + pow(x[i], penal) * (Emax - Emin);  // LinearElasticity.cc:519
}
__global__ __launch_bounds__(BLK) void k_mask_from_N(const double *__restrict__ N, uint8_t *__restrict__ mask, long nn) {
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < nn; i += (long)gridDim.x * BLK) {
        unsigned m = 0;
        for (int c = 0; c < 3; c++)
            if (N[3 * i + c] == 0.0) m |= 1u << c;
        mask[i] = (uint8_t)m;
    }
}
// cantilever load case: clamp x = xmin, line load at x = xmax, z = zmin
// (LinearElasticity.cc:143-171); coordinates are integers here, so the
// reference's epsilon windows reduce to index tests.
__global__ __launch_bounds__(BLK) void k_cantilever(Geom g, double *__restrict__ N, double *__restrict__ RHS) {
    const long nn = g.nodes();
    for (long n = blockIdx.x * (long)BLK + threadIdx.x; n < nn; n += (long)gridDim.x * BLK) {
        const int i = (int)(n % g.nx), j = (int)((n / g.nx) % g.ny), k = (int)(n / g.plane()) + g.gz0;
        const double nv = i == 0 ? 0.0 : 1.0;
        double load = 0.0;
        if (i == g.nx - 1 && k == 0) load = (j == 0 || j == g.ny - 1) ? -0.001 / 2.0 : -0.001;
        N[3 * n] = N[3 * n + 1] = N[3 * n + 2] = nv;
        RHS[3 * n] = RHS[3 * n + 1] = 0.0;
        RHS[3 * n + 2] = load;
    }
}

extern "C" int tp_elasticity_create(tp_elasticity **out, tp_grid *g, const tp_solver_opts *o) {
    return tp_elasticity_create_ke(out, g, o, nullptr);
}
// ke_host_576 != NULL: use this element matrix instead of Hex8Isoparametric's (a host that assembles through
// MatSetValuesLocal hands over the matrix it computed itself, LinearElasticity.cc:118-123, :519-524)
extern "C" int tp_elasticity_create_ke(tp_elasticity **out, tp_grid *g, const tp_solver_opts *o, const double *ke_host_576) {
    if (!out || !g || !o) return TP_ERR_ARG;
    if (o->nlvls < 1 || o->nlvls > TP_MAX_LEVELS) return TP_ERR_ARG;
    // TopOpt.cc:183-201: every direction divisible by 2^(nlvls-1); here also per slab
    const int f = 1 << (o->nlvls - 1);
    if (g->ex % f || g->ey % f || g->ez_own % f) return TP_ERR_ARG;
    if (o->ksp_mode != 0 && o->ksp_mode != 1) return TP_ERR_ARG;
    if (o->ksp_mode == 1 && g->has_comm) {
        fprintf(stderr, "topopt_amd: ksp_mode 1 (the reference's FGMRES / GMRES / SOR configuration) runs on one device only\n");
        return TP_ERR_ARG;
    }
    // slabs: every level that stays distributed keeps at least two element layers per rank (the boundary-first halo
    // overlap and the level-1 ghost rows assume a rank's two boundary planes are distinct); the replicated coarsest
    // level (three or more levels) may come down to one.  Decided from global sizes: the same answer on every rank.
    if (g->has_comm) {
        const int last_distributed = o->nlvls >= 3 ? o->nlvls - 2 : o->nlvls - 1;
        if ((g->ez_own >> last_distributed) < 2) {
            fprintf(stderr, "topopt_amd: %d element layers per rank are too few for %d multigrid levels on slabs (need >= %d)\n",
                    g->ez_own, o->nlvls, 2 << last_distributed);
            return TP_ERR_ARG;
        }
    }
    tp_elasticity *e = new tp_elasticity();
    e->grid = g;
    e->mg.grid = g;
    e->mg.nlv = o->nlvls;
    e->mg.opt = *o;
    // the replicated copy of the coarsest level is a stored stencil: levels >= 2 are; with two levels the coarsest one is
    // level 1, applied from the fine moduli (no stencil to gather) -- it stays distributed
    e->mg.allow_replicate = o->nlvls >= 3;
    e->have_bc = e->assembled = false;
    e->d_flagged = nullptr;
    e->nflagged = 0;
    e->d_colmask = nullptr;
    e->d_flag_all = e->d_corr_nodes = e->d_corr_adj = nullptr;
    e->d_dK = e->d_corr = e->d_corr_tmp = nullptr;
    e->d_KelF = nullptr;
    e->d_fidx1 = nullptr;
    e->d_M2 = nullptr;
    e->d_flag2 = nullptr;
    e->d_list2 = nullptr;
    e->nlist2 = 0;
    e->nx_first = 0;
    e->nflag_all = 0;
    if (ke_host_576) std::memcpy(e->KE, ke_host_576, sizeof(e->KE));
    else hex8_stiffness_box(g->o.hx, g->o.hy, g->o.hz, o->nu, e->KE);
    std::vector<double> M(8 * 576);
    host_child_matrices(e->KE, M.data());
    TP_TRY(e->mg.alloc_levels());
    Geom q = make_geom(g, 0);
    TP_HIP(hipMalloc((void **)&e->d_KE, sizeof(double) * 576));
    TP_HIP(hipMalloc((void **)&e->d_M, sizeof(double) * 8 * 576));
    TP_HIP(hipMalloc((void **)&e->d_E, sizeof(double) * (size_t)((long)q.ex * q.ey * (q.ez_own + 2))));
    TP_HIP(hipMemset(e->d_E, 0, sizeof(double) * (size_t)((long)q.ex * q.ey * (q.ez_own + 2))));
    TP_HIP(hipMalloc((void **)&e->d_mask, (size_t)q.nodes()));
    TP_HIP(hipMalloc((void **)&e->d_bN, sizeof(double) * 3 * (size_t)q.nodes()));
    TP_HIP(hipMalloc((void **)&e->d_N, sizeof(double) * 3 * (size_t)q.nodes()));
    TP_HIP(hipMemcpy(e->d_KE, e->KE, sizeof(double) * 576, hipMemcpyHostToDevice));
    TP_HIP(hipMemcpy(e->d_M, M.data(), sizeof(double) * 8 * 576, hipMemcpyHostToDevice));
    {
        std::vector<double> M2((size_t)64 * 576);
        host_grandchild_matrices(M.data(), M2.data());
        TP_HIP(hipMalloc((void **)&e->d_M2, sizeof(double) * M2.size()));
        TP_HIP(hipMemcpy(e->d_M2, M2.data(), sizeof(double) * M2.size(), hipMemcpyHostToDevice));
    }
    TP_HIP(hipMemset(e->d_mask, 0, (size_t)q.nodes()));
    for (int l = 0; l < e->mg.nlv; l++) {
        Level<3> &L = e->mg.lv[l];
        L.kind = l == 0 ? LV_MATFREE : LV_DIA;
        L.KE = e->d_KE;
        L.E = e->d_E;
        L.mask = e->d_mask;
        L.S = L.Kel = nullptr;
        if (l == 0) {
            // tuned kernel needs the reflection symmetry of a box element (always true here)
            SymKE sk;
            const double asym = make_sym_ke(e->KE, &sk);
            L.sym_slot = (asym < 1e-12 && !getenv("TP_NO_TILE")) ? sym_slot_acquire(sk) : -1;
            TP_HIP(hipMalloc((void **)&e->d_colmask, (size_t)q.plane()));
            TP_HIP(hipMemset(e->d_colmask, 0, (size_t)q.plane()));
            L.colmask = e->d_colmask;
            L.use_tile = L.sym_slot >= 0;
        }
        if (l == 1 && e->mg.lv[0].use_tile && !getenv("TP_NO_MACRO") && o->ksp_mode == 0) {  // Gauss-Seidel needs rows
            // level 1 is applied from the fine densities (k_matfree_tile<.,1>): no stencil storage
            L.kind = LV_MACRO;
            L.use_tile = true;
            L.sym_slot = e->mg.lv[0].sym_slot;
            {
                double gv[MACG_N];
                const double dropped = make_macro_tensor(M.data(), gv);
                if (dropped > 1e-12 || macro_slot_upload(L.sym_slot, gv)) return TP_ERR_STATE;
            }
            L.fex = q.ex;
            L.fey = q.ey;
            TP_HIP(hipMalloc((void **)&e->d_corr, sizeof(double) * (size_t)L.ndof()));
            TP_HIP(hipMemset(e->d_corr, 0, sizeof(double) * (size_t)L.ndof()));
            L.corr = e->d_corr;
        }
        if (l > 0) {
            if (L.kind == LV_DIA) {
                // 81 diagonals + 3 slices for the row-sum correction of the mirrored reads (operators.h: k_dia_sym_fix)
                TP_HIP(hipMalloc((void **)&L.S, sizeof(double) * 84 * (size_t)L.ndof()));
                TP_HIP(hipMemset(L.S, 0, sizeof(double) * 84 * (size_t)L.ndof()));
            }
            // the matrix-free level 1 materialises no element matrices (1.2 GB at 128^3)
            if (L.kind != LV_MACRO) TP_HIP(hipMalloc((void **)&L.Kel, sizeof(double) * 576 * (size_t)L.g.elems_stored()));
        }
    }
    *out = e;
    return TP_OK;
}
extern "C" int tp_elasticity_destroy(tp_elasticity *e) {
    if (!e) return TP_OK;
    (void)hipStreamSynchronize(e->grid->stream);
    sym_slot_release(e->mg.lv[0].sym_slot);
    e->mg.free_levels();
    if (e->aux_stream) (void)hipStreamDestroy(e->aux_stream);
    if (e->aux_fork) (void)hipEventDestroy(e->aux_fork);
    if (e->aux_done) (void)hipEventDestroy(e->aux_done);
    for (void *p : {(void *)e->d_KE, (void *)e->d_M, (void *)e->d_E, (void *)e->d_mask, (void *)e->d_bN, (void *)e->d_N,
                    (void *)e->d_flagged, (void *)e->d_colmask, (void *)e->d_flag_all, (void *)e->d_corr_nodes,
                    (void *)e->d_corr_adj, (void *)e->d_dK, (void *)e->d_corr, (void *)e->d_corr_tmp, (void *)e->d_KelF,
                    (void *)e->d_fidx1, (void *)e->d_M2, (void *)e->d_flag2, (void *)e->d_list2})
        (void)hipFree(p);
    delete e;
    return TP_OK;
}
extern "C" int tp_elasticity_get_ke(const tp_elasticity *e, double *ke) {
    std::memcpy(ke, e->KE, sizeof(e->KE));
    return TP_OK;
}
// The element matrix the fine-level tile kernels APPLY: KE_eff = T D T with D = the packed block-diagonal part of
// T KE T / 64 (matfree_tile.h: make_sym_ke; T = 8 x 8 Walsh-Hadamard over the element's nodes, per component).  It differs
// from KE by what the packing drops or averages -- entries of relative size ~1e-17 that are rounding residue of KE itself
// (an exactly box-symmetric KE has exact zeros there) -- so |KE_eff - KE| is below one unit in the last place of the
// largest entry.  Returned as a double-double pair (hi + lo, summed in long double on the host): the parity checks feed it to
// the 80-bit arbiter, which must see the operator the kernels see, not its rounding to double.  Without the tile kernels
// (TP_NO_TILE, a KE that is not box symmetric) the kernels apply KE itself: hi = KE, lo = 0.
static int ke_applied(const tp_elasticity *e, bool krylov, double *hi, double *lo) {
    if (!e || !hi || !lo) return TP_ERR_ARG;
    if (!e->mg.lv[0].use_tile) {
        for (int i = 0; i < 576; i++) hi[i] = e->KE[i], lo[i] = 0.0;
        return TP_OK;
    }
    SymKE sk;
    (void)make_sym_ke(e->KE, &sk);
    long double D[24][24];
    for (int i = 0; i < 24; i++)
        for (int j = 0; j < 24; j++) D[i][j] = 0.0L;
    for (int q = 0; q < 8; q++)
        for (int r = 0; r < 3; r++)
            for (int s2 = r; s2 < 3; s2++) {
                const int id = symke_idx(q, r, s2);
                if (id < 0) continue;
                const int i = (q ^ (1 << r)) * 3 + r, j = (q ^ (1 << s2)) * 3 + s2;
                D[i][j] = D[j][i] = (long double)sk.a[id];
            }
    if (krylov && SYMKE_KRYLOV) {   // the plain products also apply the translation mode's column and row (one-sided)
        for (int p2 = 0; p2 < 8; p2++)
            for (int r = 0; r < 3; r++)
                for (int s2 = 0; s2 < 3; s2++)
                    if (symx_col(p2, r, s2)) D[p2 * 3 + r][s2] = (long double)sk.x[(p2 * 3 + r) * 3 + s2];
        for (int r = 0; r < 3; r++)
            for (int p2 = 1; p2 < 8; p2++)
                for (int s2 = 0; s2 < 3; s2++) D[r][p2 * 3 + s2] = (long double)sk.x[SYMKE_XCOL + (r * 7 + p2 - 1) * 3 + s2];
    }
    for (int m = 0; m < 8; m++)
        for (int r = 0; r < 3; r++)
            for (int m2 = 0; m2 < 8; m2++)
                for (int s2 = 0; s2 < 3; s2++) {
                    long double acc = 0.0L;
                    for (int p2 = 0; p2 < 8; p2++)
                        for (int p3 = 0; p3 < 8; p3++) {
                            const int sg = (__builtin_popcount(p2 & m) + __builtin_popcount(p3 & m2)) & 1;
                            acc += sg ? -D[p2 * 3 + r][p3 * 3 + s2] : D[p2 * 3 + r][p3 * 3 + s2];
                        }
                    const int i = (3 * h_M2A[m] + r) * 24 + 3 * h_M2A[m2] + s2;
                    hi[i] = (double)acc;
                    lo[i] = (double)(acc - (long double)hi[i]);
                }
    return TP_OK;
}
extern "C" int tp_elasticity_get_ke_effective(const tp_elasticity *e, double *hi, double *lo) { return ke_applied(e, false, hi, lo); }
// The element matrix of the KRYLOV operator (the plain products A p, A x0, MatMult): KE_eff plus the translation mode's column
// and row of T KE T / 64 as KE has them (matfree_tile.h: SYMKE_KRYLOV).  Same double-double convention.
extern "C" int tp_elasticity_get_ke_krylov(const tp_elasticity *e, double *hi, double *lo) { return ke_applied(e, true, hi, lo); }
extern "C" int tp_elasticity_set_bc(tp_elasticity *e, const double *N) {
    if (e) e->mg.topology_epoch++;  // captured launch chains reference the lists rebuilt below
    tp_grid *g = e->grid;
    Geom q = make_geom(g, 0);
    const long nn = q.nodes();
    TP_LAUNCH(k_mask_from_N, dim3(grid_for(nn)), dim3(BLK), 0, g->stream, N, e->d_mask, nn);
    TP_HIP(hipMemcpyAsync(e->d_N, N, sizeof(double) * 3 * (size_t)nn, hipMemcpyDeviceToDevice, g->stream));
    e->h_mask.resize((size_t)nn);
    TP_HIP(hipMemcpyAsync(e->h_mask.data(), e->d_mask, (size_t)nn, hipMemcpyDeviceToHost, g->stream));
    TP_HIP(hipStreamSynchronize(g->stream));
    {
        std::vector<uint8_t> cm((size_t)q.plane(), 0);
        for (long n = 0; n < nn; n++) cm[(size_t)(n % q.plane())] |= e->h_mask[(size_t)n];
        TP_HIP(hipMemcpy(e->d_colmask, cm.data(), cm.size(), hipMemcpyHostToDevice));
    }
    // level-1 elements whose 27 fine nodes include a clamped one take the generic Galerkin path
    std::vector<int> fl;
    if (e->mg.nlv > 1) {
        Geom c = make_geom(g, 1);
        for (int K = 0; K < c.ez_own; K++)
            for (int J = 0; J < c.ey; J++)
                for (int I = 0; I < c.ex; I++) {
                    bool any = false;
                    for (int dk = 0; dk <= 2 && !any; dk++)
                        for (int dj = 0; dj <= 2 && !any; dj++)
                            for (int di = 0; di <= 2 && !any; di++)
                                any = e->h_mask[(size_t)((2 * I + di) + (long)q.nx * ((2 * J + dj) + (long)q.ny * (2 * K + dk)))] != 0;
                    if (any) fl.push_back(I + c.ex * (J + c.ey * K));
                }
    }
    (void)hipFree(e->d_flagged);
    e->d_flagged = nullptr;
    e->nflagged = (int)fl.size();
    if (e->mg.nlv > 1 && e->mg.lv[1].kind == LV_MACRO) {
        // matrix-free level 1: flagged elements incl. the ghost layer above (flags of the upper
        // neighbour's first layer travel as doubles through the halo mechanism)
        Level<3> &L1 = e->mg.lv[1];
        Geom c = L1.g;
        const long clay = (long)c.ex * c.ey;
        std::vector<double> fl_d((size_t)c.elems_stored(), 0.0);
        for (int id : fl) fl_d[(size_t)id] = 1.0;
        if (g->has_comm) {
            double *tmp;
            TP_HIP(hipMalloc((void **)&tmp, sizeof(double) * fl_d.size()));
            TP_HIP(hipMemcpy(tmp, fl_d.data(), sizeof(double) * fl_d.size(), hipMemcpyHostToDevice));
            TP_TRY(exchange_segments(g, tmp, nullptr, nullptr, tmp + clay * c.ez_own, clay, 1, clay));
            TP_HIP(hipStreamSynchronize(g->stream));
            TP_HIP(hipMemcpy(fl_d.data(), tmp, sizeof(double) * fl_d.size(), hipMemcpyDeviceToHost));
            (void)hipFree(tmp);
        }
        std::vector<int> fall, fidx((size_t)c.elems_stored(), -1);
        for (size_t i = 0; i < fl_d.size(); i++)
            if (fl_d[i] != 0.0) {
                fidx[i] = (int)fall.size();
                fall.push_back((int)i);
            }
        std::vector<int> cn, cadj;
        for (int k = c.own_lo; k <= c.own_hi; k++)
            for (int j = 0; j < c.ny; j++)
                for (int i = 0; i < c.nx; i++) {
                    int adj[8];
                    bool any = false;
                    for (int I = 0; I < 8; I++) {
                        const int ei = i - h_LX[I], ej = j - h_LY[I], ek = k - h_LZ[I];
                        adj[I] = -1;
                        if (ei < 0 || ei >= c.ex || ej < 0 || ej >= c.ey || ek < 0 || ek >= c.ezl) continue;
                        adj[I] = fidx[(size_t)(ei + (long)c.ex * (ej + (long)c.ey * ek))];
                        any = any || adj[I] >= 0;
                    }
                    if (any) {
                        cn.push_back((int)(i + (long)c.nx * (j + (long)c.ny * k)));
                        cadj.insert(cadj.end(), adj, adj + 8);
                    }
                }
        for (void *p : {(void *)e->d_flag_all, (void *)e->d_corr_nodes, (void *)e->d_corr_adj, (void *)e->d_dK,
                        (void *)e->d_corr_tmp, (void *)e->d_KelF, (void *)e->d_fidx1})
            (void)hipFree(p);
        e->d_flag_all = e->d_corr_nodes = e->d_corr_adj = nullptr;
        e->d_dK = e->d_corr_tmp = e->d_KelF = nullptr;
        e->d_fidx1 = nullptr;
        // own flagged elements come first in `fall` (same order as `fl`), the ghost layer last; the first own layer
        // is the head of `fl`.  The compact ghost rows travel down with a common row count (max over the ranks).
        int n0 = 0;
        for (int id : fl) n0 += id < clay ? 1 : 0;
        e->nx_first = n0;
        if (g->has_comm) {
            std::vector<double> cnt((size_t)g->nranks, 0.0);
            cnt[(size_t)g->rank] = n0;
            for (int o = 0; o < g->nranks; o += 16) {
                const int m16 = g->nranks - o < 16 ? g->nranks - o : 16;
                TP_HIP(hipMemcpy(g->comm.red, cnt.data() + o, sizeof(double) * m16, hipMemcpyHostToDevice));
                if (g->comm.allreduce_sum(g->comm.user, m16)) return TP_ERR_COMM;
                TP_HIP(hipStreamSynchronize(g->stream));
                TP_HIP(hipMemcpy(cnt.data() + o, g->comm.red, sizeof(double) * m16, hipMemcpyDeviceToHost));
            }
            for (double v : cnt) e->nx_first = v > e->nx_first ? (int)v : e->nx_first;
        }
        TP_HIP(hipMalloc((void **)&e->d_fidx1, sizeof(int) * fidx.size()));
        TP_HIP(hipMemcpy(e->d_fidx1, fidx.data(), sizeof(int) * fidx.size(), hipMemcpyHostToDevice));
        (void)hipFree(e->d_flag2);
        (void)hipFree(e->d_list2);
        e->d_flag2 = nullptr;
        e->d_list2 = nullptr;
        e->nlist2 = 0;
        if (e->mg.nlv > 2) {  // own level-2 elements with a flagged child take the generic construction
            Geom c2 = make_geom(g, 2);
            std::vector<uint8_t> f2((size_t)c2.own_elems(), 0);
            std::vector<int> l2;
            for (int K = 0; K < c2.ez_own; K++)
                for (int J = 0; J < c2.ey; J++)
                    for (int I = 0; I < c2.ex; I++) {
                        bool any = false;
                        for (int ch = 0; ch < 8 && !any; ch++)
                            any = fidx[(size_t)((2 * I + (ch & 1)) + (long)c.ex * ((2 * J + ((ch >> 1) & 1)) + (long)c.ey * (2 * K + (ch >> 2))))] >= 0;
                        if (any) {
                            f2[(size_t)(I + (long)c2.ex * (J + (long)c2.ey * K))] = 1;
                            l2.push_back(I + c2.ex * (J + c2.ey * K));
                        }
                    }
            TP_HIP(hipMalloc((void **)&e->d_flag2, f2.size() + 1));
            TP_HIP(hipMemcpy(e->d_flag2, f2.data(), f2.size(), hipMemcpyHostToDevice));
            e->nlist2 = (int)l2.size();
            if (!l2.empty()) {
                TP_HIP(hipMalloc((void **)&e->d_list2, sizeof(int) * l2.size()));
                TP_HIP(hipMemcpy(e->d_list2, l2.data(), sizeof(int) * l2.size(), hipMemcpyHostToDevice));
            }
        }
        {
            const size_t rows = fl.size() + 2 * (size_t)e->nx_first + 1;
            TP_HIP(hipMalloc((void **)&e->d_KelF, sizeof(double) * 576 * rows));
            TP_HIP(hipMemset(e->d_KelF, 0, sizeof(double) * 576 * rows));
        }
        e->nflag_all = (int)fall.size();
        TP_HIP(hipMemset(e->d_corr, 0, sizeof(double) * (size_t)L1.ndof()));
        if (!fall.empty()) {
            TP_HIP(hipMalloc((void **)&e->d_flag_all, sizeof(int) * fall.size()));
            TP_HIP(hipMemcpy(e->d_flag_all, fall.data(), sizeof(int) * fall.size(), hipMemcpyHostToDevice));
            TP_HIP(hipMalloc((void **)&e->d_dK, sizeof(double) * 576 * fall.size()));
            TP_HIP(hipMalloc((void **)&e->d_corr_tmp, sizeof(double) * 24 * fall.size()));
        }
        if (!cn.empty()) {
            TP_HIP(hipMalloc((void **)&e->d_corr_nodes, sizeof(int) * cn.size()));
            TP_HIP(hipMemcpy(e->d_corr_nodes, cn.data(), sizeof(int) * cn.size(), hipMemcpyHostToDevice));
            TP_HIP(hipMalloc((void **)&e->d_corr_adj, sizeof(int) * cadj.size()));
            TP_HIP(hipMemcpy(e->d_corr_adj, cadj.data(), sizeof(int) * cadj.size(), hipMemcpyHostToDevice));
        }
        L1.dK = e->d_dK;
        L1.flag_list = e->d_flag_all;
        L1.corr_nodes = e->d_corr_nodes;
        L1.corr_adj = e->d_corr_adj;
        L1.ncorr_nodes = (int)cn.size();
        L1.nflag = e->nflag_all;
        L1.corr_tmp = e->d_corr_tmp;
    }
    if (e->nflagged) {
        TP_HIP(hipMalloc((void **)&e->d_flagged, sizeof(int) * fl.size()));
        TP_HIP(hipMemcpy(e->d_flagged, fl.data(), sizeof(int) * fl.size(), hipMemcpyHostToDevice));
    }
    e->have_bc = true;
    e->assembled = false;
    return TP_OK;
}
extern "C" int tp_elasticity_cantilever(tp_elasticity *e, double *N, double *RHS) {
    tp_grid *g = e->grid;
    Geom q = make_geom(g, 0);
    TP_LAUNCH(k_cantilever, dim3(grid_for(q.nodes())), dim3(BLK), 0, g->stream, q, N, RHS);
    return tp_elasticity_set_bc(e, N);
}

static int elasticity_setup_from_E(tp_elasticity *e);
// Did ANY rank raise `mine`?  The give-up of a one-XCD kernel is seen by the rank it happened on only, but what follows
// from it -- the hierarchy rebuilt, the solve repeated, the one-XCD forms off from then on -- contains halo exchanges and
// reductions and changes the (replicated) coarse solve: every rank of the grid must take the same branch.
// One rank: no communication.  Collective over the grid's communicator otherwise: callers reach it on every rank or on none.
static int agree_any(tp_grid *g, bool mine, bool *any) {
    *any = mine;
    if (!g->has_comm) return TP_OK;
    double v = mine ? 1.0 : 0.0;
    TP_HIP(hipMemcpyAsync(g->comm.red, &v, sizeof(double), hipMemcpyHostToDevice, g->stream));
    TP_HIP(hipStreamSynchronize(g->stream));  // (v lives on this stack frame)
    if (g->comm.allreduce_sum(g->comm.user, 1)) return TP_ERR_COMM;
    TP_HIP(hipMemcpyAsync(&v, g->comm.red, sizeof(double), hipMemcpyDeviceToHost, g->stream));
    TP_HIP(hipStreamSynchronize(g->stream));
    *any = v > 0.0;
    return TP_OK;
}
// The same agreement with a severity: 0 none, 1 soft (only the DEFERRED factorisation gave up: stop deferring), 2 hard
// (one-XCD forms off).  The ranks take the hardest one any of them saw (sum of 1 / 4096 per rank: no rank count reaches 4096).
static int agree_giveup(tp_grid *g, int mine, int *agreed) {
    *agreed = mine;
    if (!g->has_comm) return TP_OK;
    double v = mine == 2 ? 4096.0 : (mine == 1 ? 1.0 : 0.0);
    TP_HIP(hipMemcpyAsync(g->comm.red, &v, sizeof(double), hipMemcpyHostToDevice, g->stream));
    TP_HIP(hipStreamSynchronize(g->stream));
    if (g->comm.allreduce_sum(g->comm.user, 1)) return TP_ERR_COMM;
    TP_HIP(hipMemcpyAsync(&v, g->comm.red, sizeof(double), hipMemcpyDeviceToHost, g->stream));
    TP_HIP(hipStreamSynchronize(g->stream));
    *agreed = v >= 4096.0 ? 2 : (v > 0.0 ? 1 : 0);
    return TP_OK;
}
// A one-XCD persistent kernel gave up (common.h: tp_xcd_disabled).  Hard: switch those forms off for the rest of the process.
// Soft (round 6): only the factorisation chain gave up, and it ran deferred beside the head of the solve -- keep the
// one-XCD forms, join the factorisation at the end of the set-up from now on (what TP_NO_DEFER_FACTOR does).  Either way hand the
// control blocks back zeroed and build the hierarchy again from the moduli.
static int redo_without_xcd(tp_elasticity *e, const char *where, bool soft = false) {
    tp_giveup_count()++;
    if (soft) {
        fprintf(stderr, "topopt_amd: the coarse factorisation (one-XCD kernels, deferred beside the head of the solve) gave up during %s; "
                        "redoing it, the factorisation is joined at the end of the set-up from now on (one-XCD forms stay on)\n", where);
        tp_defer_disabled() = true;
    } else {
        fprintf(stderr, "topopt_amd: a one-XCD persistent kernel gave up during %s (workgroups not co-resident: shared device?); "
                        "redoing it with separate launches, one-XCD forms off for the rest of this process\n", where);
        tp_xcd_disabled() = true;
    }
    e->mg.join_side_streams();
    e->mg.xcd_reset_controls();
    TP_HIP(hipStreamSynchronize(e->grid->stream));
    return elasticity_setup_from_E(e);
}
// give-ups recovered from so far in this process, and what they switched off
extern "C" int tp_xcd_status(int *giveups, int *xcd_forms_off, int *defer_off) {
    if (giveups) *giveups = tp_giveup_count();
    if (xcd_forms_off) *xcd_forms_off = tp_xcd_disabled() ? 1 : 0;
    if (defer_off) *defer_off = tp_defer_disabled() ? 1 : 0;
    return TP_OK;
}
extern "C" int tp_elasticity_assemble(tp_elasticity *e, const double *xPhys, double Emin, double Emax, double penal) {
    if (!e->have_bc) return TP_ERR_STATE;
    tp_grid *g = e->grid;
    MGSolver<3> &mg = e->mg;
    hipStream_t s = g->stream;
    Geom q0 = mg.lv[0].g;
    const long nel = q0.own_elems(), lay = (long)q0.ex * q0.ey;
    TP_LAUNCH(k_simp, dim3(grid_for(nel)), dim3(BLK), 0, s, xPhys, Emin, Emax, penal, e->d_E, nel);
    count_launch(g, 16.0 * nel, 3.0 * nel);
    // two ghost layers above <- upper neighbour's first own layers (level 1 is applied
    // from the fine densities and reaches one coarse = two fine layers up)
    TP_TRY(exchange_segments(g, e->d_E, nullptr, nullptr, e->d_E + nel, 2 * lay, 1, 2 * lay));
    // a previous assembly that failed half way must not leave its state behind (a stale "factorisation under way" event,
    // a sticky give-up flag)
    mg.cd_early = mg.cd_inverse_owed = false;
    for (bool &b : mg.lv_ready_set) b = false;
    e->assembled = false;
    TP_TRY(mg.join_pending_factor());  // a factorisation no solve has waited for must not be overtaken by the new coarse stencil
    int rc = elasticity_setup_from_E(e);
    if (rc == TP_OK && mg.opt.ksp_mode == 0 && !tp_xcd_disabled()) {
        // a Lanczos run on one XCD that gave up poisons its Ritz values with NaN (a factorisation that gave up shows in
        // the solve: tp_elasticity_solve)
        bool bad = false;
        // the levels, and the slots of their replicated copies (mg.h: rix -- slot nlv + (l - rep0) holds the copy of level l >= rep0)
        const int nslots = mg.nlv + (mg.replicate ? mg.nlv - mg.rep0 : 0);
        for (int l = 0; l < nslots && l < MGSolver<3>::LV_SLOTS; l++) bad = bad || mg.lv[l].lam != mg.lv[l].lam || mg.lv[l].lam_min != mg.lv[l].lam_min;
        // TP_TEST_FORCE_GIVEUP=2 / =1 (tests/test_gpu_parity.py::test_one_xcd_kernels_give_up_path): take the recovery branch of
        // the set-up / of the solve once without a real give-up; "=2:R" / "=1:R" on rank R only (the multi-rank agreement)
        static const bool force = tp_test_force_giveup(2, g->rank);
        // several ranks: the decision is taken together, and only where a one-XCD Lanczos run exists on this grid at all
        // (lan_ctl is allocated by the first one; same configuration on every rank) -- the default set-up pays nothing
        bool mine = (bad && mg.xcd_gaveup()) || (force && !tp_xcd_disabled()), any = mine;
        if (g->has_comm && (mg.lan_ctl != nullptr || tp_test_force_giveup(2, -1))) rc = agree_any(g, mine, &any);
        if (rc == TP_OK && any) rc = redo_without_xcd(e, "the set-up");
    }
    if (rc != TP_OK) {  // join what the failed set-up left running on the side streams
        if (e->aux_stream) (void)hipStreamSynchronize(e->aux_stream);
        mg.join_side_streams();
        (void)hipStreamSynchronize(s);
    }
    return rc;
}
static int elasticity_setup_from_E(tp_elasticity *e) {
    tp_grid *g = e->grid;
    MGSolver<3> &mg = e->mg;
    hipStream_t s = g->stream;
    Geom q0 = mg.lv[0].g;
    const long nel = q0.own_elems();
    const bool macro1 = mg.nlv > 1 && mg.lv[1].kind == LV_MACRO;  // level 1 applied from E: no element matrices there
    // Order of the set-up (round 3): the chain fine moduli -> element matrices of level 2 -> ... -> stencil of the coarsest
    // level is what the factorisation of that level waits for (coarse_direct.h: the longest chain of the set-up), so it
    // runs first and bare (pass 1: the Galerkin kernels and their ghost exchanges only); everything that merely FINISHES a
    // level -- stencils of the middle levels, level 1's corrections and diagonal, the fine diagonal -- follows (pass 2)
    // while the factorisation is already under way on its own stream.  Same kernels, same data, same results.
    // Level 2's element matrices come straight from the fine moduli (k_galerkin_l2_fast, ~0.3 ms at 128^3) and do not
    // depend on level 1's kernels: side by side on a second stream, joined where level 2 continues.
    static const bool no_l2_aside = getenv("TP_NO_L2_ASIDE") != nullptr || getenv("TP_NO_L2_FAST") != nullptr || tp_debug_sync();
    static const bool two_pass = getenv("TP_NO_SETUP_REORDER") == nullptr;
    bool l2_aside = false;
    if (macro1 && mg.nlv > 2 && !no_l2_aside) {
        if (!e->aux_stream) TP_HIP(hipStreamCreateWithFlags(&e->aux_stream, hipStreamNonBlocking));
        if (!e->aux_fork) TP_HIP(hipEventCreateWithFlags(&e->aux_fork, hipEventDisableTiming));
        if (!e->aux_done) TP_HIP(hipEventCreateWithFlags(&e->aux_done, hipEventDisableTiming));
        Level<3> &C2 = mg.lv[2];
        const long nEc2 = C2.g.own_elems();
        static const long nb2_env = getenv("TP_L2_BLOCKS") ? atol(getenv("TP_L2_BLOCKS")) : 512;
        const unsigned nb2 = (unsigned)(nEc2 < nb2_env ? nEc2 : nb2_env);
        TP_HIP(hipEventRecord(e->aux_fork, s));
        TP_HIP(hipStreamWaitEvent(e->aux_stream, e->aux_fork, 0));
        TP_LAUNCH(k_galerkin_l2_fast, dim3(nb2, 3), dim3(L2F_T), 0, e->aux_stream, mg.lv[0].g, C2.g, e->d_E, e->d_M2, C2.Kel, (int)nEc2);
        count_launch(g, 8.0 * 64 * nEc2 + 8.0 * 576 * nEc2, 2.0 * 64 * 576 * nEc2);
        TP_HIP(hipEventRecord(e->aux_done, e->aux_stream));
        l2_aside = true;
    }
    // ---- pass 1 of level l: its Galerkin element matrices (and their ghost layer)
    auto galerkin_level = [&](int l) -> int {
        Level<3> &F = mg.lv[l - 1], &C = mg.lv[l];
        const long nEc = C.g.own_elems();
        if (l == 1 && macro1) {
            // only the flagged elements get their (exact, masked) Galerkin matrix: compact rows, own ones first,
            // then the ghost layer's -- the upper neighbour's first-layer rows, which head ITS array
            if (e->nflagged) {
                TP_LAUNCH(k_galerkin_fine_masked, dim3(e->nflagged), dim3(64), 0, s, F.g, C.g, e->d_E, e->d_KE,
                                   e->d_mask, e->d_flagged, e->d_KelF, 1);
                count_launch(g);
            }
            if (g->has_comm && e->nx_first > 0)
                TP_TRY(exchange_segments(g, e->d_KelF, nullptr, nullptr, e->d_KelF + 576 * (long)e->nflagged, 576,
                                         e->nx_first, 576));
            return TP_OK;
        }
        if (l == 1) {
            TP_LAUNCH(k_galerkin_fine_fast, dim3((unsigned)nEc), dim3(192), 0, s, F.g, C.g, e->d_E, e->d_M,
                               C.Kel);
            count_launch(g, 8.0 * nel + 8.0 * 576 * nEc, 2.0 * 8 * 576 * nEc);
            if (e->nflagged) {
                TP_LAUNCH(k_galerkin_fine_masked, dim3(e->nflagged), dim3(64), 0, s, F.g, C.g, e->d_E, e->d_KE,
                                   e->d_mask, e->d_flagged, C.Kel, 0);
                count_launch(g);
            }
        } else if (l == 2 && macro1) {
            // all 64 fine moduli below an element at once (constants in registers); elements with a flagged
            // level-1 child go through the generic contraction with the compact rows
            static const bool no_fast2 = getenv("TP_NO_L2_FAST") != nullptr;
            if (!no_fast2) {
                static const long nb2_env = getenv("TP_L2_BLOCKS") ? atol(getenv("TP_L2_BLOCKS")) : 512;
                const unsigned nb2 = (unsigned)(nEc < nb2_env ? nEc : nb2_env);
                if (l2_aside) {
                    TP_HIP(hipStreamWaitEvent(s, e->aux_done, 0));
                } else {
                    TP_LAUNCH(k_galerkin_l2_fast, dim3(nb2, 3), dim3(L2F_T), 0, s, mg.lv[0].g, C.g, e->d_E, e->d_M2,
                                       C.Kel, (int)nEc);
                    count_launch(g, 8.0 * 64 * nEc + 8.0 * 576 * nEc, 2.0 * 64 * 576 * nEc);
                }
                if (e->nlist2) {
                    TP_LAUNCH((k_galerkin_coarse<true>), dim3((unsigned)e->nlist2), dim3(64), 0, s, F.g, C.g,
                                       e->d_KelF, C.Kel, mg.lv[0].g, e->d_E, e->d_M, e->d_fidx1, (long)e->nlist2, e->d_list2);
                    count_launch(g);
                }
            } else {
                const unsigned nblk = (unsigned)(nEc < 4096 ? nEc : 4096);
                TP_LAUNCH((k_galerkin_coarse<true>), dim3(nblk), dim3(64), 0, s, F.g, C.g, e->d_KelF, C.Kel,
                                   mg.lv[0].g, e->d_E, e->d_M, e->d_fidx1, nEc, (const int *)nullptr);
                count_launch(g, 8.0 * 64 * nEc + 8.0 * 576 * nEc, 2.0 * (8 * 576 * 8 + 0.18 * 8 * 64 * 64 * 9) * nEc);
            }
        } else {
            TP_LAUNCH((k_galerkin_coarse<false>), dim3((unsigned)nEc), dim3(64), 0, s, F.g, C.g, F.Kel, C.Kel,
                               mg.lv[0].g, nullptr, nullptr, nullptr, nEc, (const int *)nullptr);
            count_launch(g, 8.0 * 576 * (9.0 * nEc), 2.0 * 0.18 * 8 * 64 * 64 * 9 * nEc);
        }
        // coarse ghost element layer above <- upper neighbour's first own layer
        const long clay = (long)C.g.ex * C.g.ey;
        // (one contiguous block of 576*clay doubles, cut into rows of `clay` so that it fits the staging buffers)
        TP_TRY(exchange_segments(g, C.Kel, nullptr, nullptr, C.Kel + 576 * clay * C.g.ez_own, clay, 576, clay));
        return TP_OK;
    };
    // ---- pass 2 of level l: what the level's own operator needs (stencil by diagonals / corrections, Jacobi diagonal)
    auto finish_level = [&](int l) -> int {
        Level<3> &F = mg.lv[l - 1], &C = mg.lv[l];
        const int gn = (int)((C.g.owned_nodes() + BLK - 1) / BLK);
        if (l == 1 && macro1) {
            if (e->nflag_all) {
                TP_LAUNCH(k_macro_delta, dim3((int)(((long)e->nflag_all * 576 + BLK - 1) / BLK)), dim3(BLK), 0, s,
                                   F.g, C.g, e->d_E, e->d_M, e->d_KelF, e->d_flag_all, e->nflag_all, e->d_dK);
                count_launch(g);
            }
            TP_LAUNCH(k_macro_diag, dim3(gn), dim3(BLK), 0, s, F.g, C.g, e->d_E, e->d_M, e->d_KelF, e->d_fidx1,
                               C.dinv);
            count_launch(g, 8.0 * 8.0 * C.g.own_elems() + 24.0 * C.g.owned_nodes(), 8.0 * 64 * 3 * C.g.owned_nodes());
            return TP_OK;
        }
        if (C.kind == LV_MACRO) {  // (not reached: handled above)
            TP_LAUNCH(k_elem_diag, dim3(gn), dim3(BLK), 0, s, C.g, C.Kel, C.dinv);
            count_launch(g, 8.0 * (24.0 + 3.0) * C.g.owned_nodes(), 24.0 * C.g.owned_nodes());
        } else {
            TP_LAUNCH(k_elem_to_dia, dim3(gn, 27), dim3(BLK), 0, s, C.g, C.Kel, C.S, C.dinv);
            count_launch(g, 8.0 * (576.0 * C.g.elems_stored() + 243.0 * C.g.owned_nodes()), 9.0 * 64 * C.g.owned_nodes());
            TP_LAUNCH(k_dia_sym_fix, dim3(gn), dim3(BLK), 0, s, C.g, C.S, (long)C.ndof());
            count_launch(g, 8.0 * 243.0 * C.g.owned_nodes(), 2.0 * 117 * C.g.owned_nodes());
        }
        return TP_OK;
    };
    if (two_pass) {
        for (int l = 1; l < mg.nlv; l++) TP_TRY(galerkin_level(l));
        bool early = false;
        if (mg.nlv > 2) {  // the coarsest level's stencil first; its factorisation starts at once (one rank)
            TP_TRY(finish_level(mg.nlv - 1));
            TP_TRY(mg.coarse_direct_early(&early));
        }
        TP_TRY(mg.setup_matfree_level(0, e->KE));
        for (int l = 1; l < mg.nlv - (mg.nlv > 2 ? 1 : 0); l++) {
            TP_TRY(finish_level(l));
            TP_TRY(mg.mark_level_ready(l));  // (its spectrum chain may start from here, mg.h)
        }
    } else {
        TP_TRY(mg.setup_matfree_level(0, e->KE));
        for (int l = 1; l < mg.nlv; l++) {
            TP_TRY(galerkin_level(l));
            TP_TRY(finish_level(l));
        }
    }
    mg.side_stream = e->aux_stream;  // (idle from here on: a second stream for the spectra chains, mg.h)
    mg.ready = true;
    TP_TRY(mg.setup_replicated());
    if (mg.opt.ksp_mode == 0) TP_TRY(mg.estimate_spectra(mg.opt.fine_eig ? 0 : 1));  // Chebyshev windows
    e->assembled = true;
    return TP_OK;
}

extern "C" int tp_elasticity_apply(tp_elasticity *e, const double *u, double *y) {
    if (!e->assembled) return TP_ERR_STATE;
    return e->mg.apply(0, const_cast<double *>(u), y);
}
// the same product with the operator of the Krylov method (tp_elasticity_get_ke_krylov): what KSPSolve's CG multiplies with
extern "C" int tp_elasticity_apply_krylov(tp_elasticity *e, const double *u, double *y) {
    if (!e || !e->assembled) return TP_ERR_STATE;
    return e->mg.apply_krylov(const_cast<double *>(u), y);
}

__global__ __launch_bounds__(BLK) void k_mul(double *__restrict__ y, const double *__restrict__ a,
                                             const double *__restrict__ b, long n) {
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) y[i] = a[i] * b[i];
}

extern "C" int tp_elasticity_solve(tp_elasticity *e, const double *RHS, double *U, int *its, double *rnorm,
                                   double *bnorm, double *hist, int hist_cap) {
    if (!e->assembled) return TP_ERR_STATE;
    tp_grid *g = e->grid;
    const long n = e->mg.lv[0].ndof();
    // RHS <- RHS .* N (LinearElasticity.cc:542), on a scratch copy
    TP_LAUNCH(k_mul, dim3(grid_for(n)), dim3(BLK), 0, g->stream, e->d_bN, RHS, e->d_N, n);
    count_launch(g, 24.0 * n, 1.0 * n);
    int rc = e->mg.solve(e->d_bN, U, its, rnorm, bnorm, hist, hist_cap);
    // A solve that "diverged" because a one-XCD kernel gave up (its result is poisoned with NaN on purpose): build the
    // hierarchy again without those kernels and solve once more, from a zero guess (U holds NaN by now).
    static const bool force = tp_test_force_giveup(1, g->rank), force_any = tp_test_force_giveup(1, -1);
    // Several ranks: a divergence is seen by all of them (the residual norm is a sum over ranks), the give-up flag behind it
    // by one -- they agree on it before any of them rebuilds (agree_any is reached on every rank: rc and the test switch are
    // the same everywhere).  The reported iteration count and history are those of the SECOND solve.
    // TP_TEST_FORCE_GIVEUP=3: the soft branch (as if the deferred factorisation alone had given up)
    static const bool force3 = tp_test_force_giveup(3, g->rank), force3_any = tp_test_force_giveup(3, -1);
    if (!tp_xcd_disabled() && (rc == TP_ERR_DIVERGED || ((force_any || (force3_any && !tp_defer_disabled())) && rc == TP_OK))) {
        int mine = 0, sev = 0;
        if (rc == TP_ERR_DIVERGED && e->mg.gaveup_seen)
            mine = (e->mg.gaveup_mask == 4 && e->mg.cd_deferred_last && !tp_defer_disabled()) ? 1 : 2;
        else if (rc == TP_OK && force)
            mine = 2;
        else if (rc == TP_OK && force3 && !tp_defer_disabled())
            mine = 1;
        TP_TRY(agree_giveup(g, mine, &sev));
        if (sev) {
            TP_TRY(redo_without_xcd(e, "the solve", sev == 1));
            TP_HIP(hipMemsetAsync(U, 0, sizeof(double) * (size_t)n, g->stream));
            rc = e->mg.solve(e->d_bN, U, its, rnorm, bnorm, hist, hist_cap);
            fprintf(stderr, "topopt_amd: the solve was repeated from a zero guess: iteration count and residual history are the second solve's\n");
        }
    }
    return rc;
}

// fx = sum_e E_e u_e^T KE u_e, dfdx_e = -p x^(p-1) (Emax-Emin) u_e^T KE u_e, partial sum x
// (LinearElasticity.cc:405-437); one thread per own element, KE rows wave-uniform.
// REDUCE = false: the sensitivities alone (LinearElasticity.cc:299-361) -- no sums, nothing for the host to wait for.
template <bool REDUCE>
__global__ __launch_bounds__(BLK) void k_objective(Geom g, const double *__restrict__ KE, const double *__restrict__ U,
                                                   const double *__restrict__ x, double Emin, double Emax, double penal,
                                                   double *__restrict__ dfdx, double *__restrict__ partials) {
    const long nel = g.own_elems();
    const long t = blockIdx.x * (long)BLK + threadIdx.x;
    double f = 0.0, vol = 0.0;
    if (t < nel) {
        const int i = (int)(t % g.ex), j = (int)((t / g.ex) % g.ey), k = (int)(t / ((long)g.ex * g.ey));
        double ue[24];
#pragma unroll
        for (int a = 0; a < 8; a++) {
            const long nd = (long)(i + LXc(a)) + (long)g.nx * ((j + LYc(a)) + (long)g.ny * (k + LZc(a)));
#pragma unroll
            for (int c = 0; c < 3; c++) ue[3 * a + c] = U[3 * nd + c];
        }
        double uKu = 0.0;
#pragma unroll
        for (int r = 0; r < 24; r++) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < 24; c++) s = fma(KE[r * 24 + c], ue[c], s);
            uKu = fma(ue[r], s, uKu);
        }
        const double xe = x[t];
        f = (Emin + pow(xe, penal) * (Emax - Emin)) * uKu;
        vol = xe;
        if (dfdx) dfdx[t] = -1.0 * penal * pow(xe, penal - 1) * (Emax - Emin) * uKu;
    }
    if (!REDUCE) return;
    f = block_sum(f);
    vol = block_sum(vol);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = f;
        partials[gridDim.x + blockIdx.x] = vol;
    }
}

extern "C" int tp_elasticity_objective(tp_elasticity *e, const double *U, const double *xPhys, double Emin, double Emax,
                                       double penal, double volfrac, double *fx, double *gx, double *dfdx,
                                       double *dgdx) {
    tp_grid *g = e->grid;
    Geom q = e->mg.lv[0].g;
    const long nel = q.own_elems();
    const long nel_glob = (long)g->ex * g->ey * g->ez_glob;
    TP_TRY(halo_nodes(g, q, const_cast<double *>(U), 3));  // DMGlobalToLocal, :388-390
    const int nb = (int)((nel + BLK - 1) / BLK);
    if (!fx && !gx) {  // sensitivities only: no reduction, no host synchronisation
        if (dfdx) {
            TP_LAUNCH(k_objective<false>, dim3(nb), dim3(BLK), 0, g->stream, q, e->d_KE, U, xPhys, Emin, Emax, penal, dfdx, g->partials);
            count_launch(g, 24.0 * q.owned_nodes() + 16.0 * nel, 2.0 * 600 * nel);
        }
    } else {
        TP_LAUNCH(k_objective<true>, dim3(nb), dim3(BLK), 0, g->stream, q, e->d_KE, U, xPhys, Emin, Emax, penal, dfdx, g->partials);
        count_launch(g, 24.0 * q.owned_nodes() + 16.0 * nel, 2.0 * 600 * nel);
        TP_TRY(reduce_partials<2>(g, nb, S_TMP));
        double v[2];
        TP_TRY(read_scal(g, S_TMP, 2, v));
        if (fx) *fx = v[0];
        if (gx) *gx = v[1] / (double)nel_glob - volfrac;
    }
    if (dgdx) TP_TRY(tp_vec_set(g, dgdx, 1.0 / (double)nel_glob, nel));
    return TP_OK;
}
// The reference's split forms (main.cc can call either pair instead of the fused method):
// ComputeObjectiveConstraints minus the solve (LinearElasticity.cc:237-294): fx and gx of the state U, no sensitivities
extern "C" int tp_elasticity_objective_only(tp_elasticity *e, const double *U, const double *xPhys, double Emin, double Emax,
                                            double penal, double volfrac, double *fx, double *gx) {
    if (!e || !U || !xPhys || !fx || !gx) return TP_ERR_ARG;
    return tp_elasticity_objective(e, U, xPhys, Emin, Emax, penal, volfrac, fx, gx, nullptr, nullptr);
}
// ComputeSensitivities (LinearElasticity.cc:299-361): dfdx, dgdx of the state U as it is -- no solve, no sums
extern "C" int tp_elasticity_sensitivities(tp_elasticity *e, const double *U, const double *xPhys, double Emin, double Emax,
                                           double penal, double *dfdx, double *dgdx) {
    if (!e || !U || !xPhys || !dfdx) return TP_ERR_ARG;
    return tp_elasticity_objective(e, U, xPhys, Emin, Emax, penal, 0.0, nullptr, nullptr, dfdx, dgdx);
}

extern "C" int tp_elasticity_set_tolerances(tp_elasticity *e, double rtol, double atol, double dtol, int max_it) {
    if (!e) return TP_ERR_ARG;   // KSPSetTolerances (LinearElasticity.cc:646); negative = keep (PETSC_DEFAULT)
    if (rtol >= 0) e->mg.opt.rtol = rtol;
    if (atol >= 0) e->mg.opt.atol = atol;
    if (dtol >= 0) e->mg.opt.dtol = dtol;
    if (max_it >= 0) e->mg.opt.max_it = max_it;
    return TP_OK;
}
// The solver configuration as a literal PETSc 3.11 option string (PETSc numbers the levels from the coarsest = 0 to
// the finest = nlvls - 1; level 0 takes the -mg_coarse_ prefix), with the numeric Chebyshev windows of the LAST
// assembly: pasted next to KSPSetFromOptions (LinearElasticity.cc:659; the level KSPs read their options in
// PCSetUp_MG, after the reference's hard-coded KSPSetType calls) it reproduces this solver inside the reference.
extern "C" int tp_elasticity_petsc_options(const tp_elasticity *e, char *buf, size_t cap) {
    if (!e || !e->assembled) return -1;
    const MGSolver<3> &mg = e->mg;
    std::string o;
    char t[512];
    if (mg.opt.ksp_mode == 1) {  // what the reference's SetUpSolver hard-codes (LinearElasticity.cc:620-746)
        const char *pcn[2] = {"jacobi", "sor"};
        snprintf(t, sizeof t,
                 "-ksp_type fgmres -ksp_gmres_restart %d -ksp_rtol %.17g -ksp_atol %.17g -ksp_divtol %.17g -ksp_max_it %d "
                 "-ksp_initial_guess_nonzero true -pc_type mg -pc_mg_levels %d -pc_mg_type multiplicative -pc_mg_cycle_type v "
                 "-pc_mg_galerkin both -mg_levels_ksp_type gmres -mg_levels_ksp_gmres_restart %d -mg_levels_ksp_max_it %d "
                 "-mg_levels_pc_type %s -mg_coarse_ksp_type gmres -mg_coarse_ksp_gmres_restart %d -mg_coarse_ksp_rtol %.17g "
                 "-mg_coarse_ksp_max_it %d -mg_coarse_pc_type %s",
                 mg.opt.restart, mg.opt.rtol, mg.opt.atol, mg.opt.dtol, mg.opt.max_it, mg.nlv, mg.opt.nsmooth, mg.opt.nsmooth,
                 pcn[mg.opt.smooth_pc ? 1 : 0], mg.opt.coarse_restart, mg.opt.coarse_rtol, mg.opt.ncoarse, pcn[mg.opt.coarse_pc ? 1 : 0]);
        o = t;
        if (buf && cap > 0) {
            const size_t n = o.size() < cap - 1 ? o.size() : cap - 1;
            memcpy(buf, o.data(), n);
            buf[n] = 0;
        }
        return (int)o.size();
    }
    snprintf(t, sizeof t,
             "-ksp_type cg -ksp_norm_type unpreconditioned -ksp_rtol %.17g -ksp_atol %.17g -ksp_divtol %.17g -ksp_max_it %d "
             "-ksp_initial_guess_nonzero true -pc_type mg -pc_mg_levels %d -pc_mg_type multiplicative -pc_mg_cycle_type %s "
             "-pc_mg_galerkin both",
             mg.opt.rtol, mg.opt.atol, mg.opt.dtol, mg.opt.max_it, mg.nlv, [&] {
                 // (per-level cycle types exist in PETSc through PCMGSetCycleTypeOnLevel only: the string says w when every
                 // level that can cycle twice does)
                 bool w = mg.nlv > 2;
                 for (int l = 0; l + 2 < mg.nlv; l++) w = w && mg.cycles[l] == 2;
                 return w ? "w" : "v";
             }());
    o += t;
    for (int l = 0; l < mg.nlv; l++) {
        const Level<3> &L = mg.lv[l];
        const int k = mg.nlv - 1 - l;  // PETSc level number
        const bool coarse = (l == mg.nlv - 1 && l > 0);
        char pre[32];
        if (k == 0 && mg.nlv > 1) snprintf(pre, sizeof pre, "mg_coarse");
        else snprintf(pre, sizeof pre, "mg_levels_%d", k);
        const double lo = coarse ? L.lam_min : mg.opt.cheb_lo * L.lam, hi = mg.opt.cheb_hi * L.lam;
        snprintf(t, sizeof t,
                 " -%s_ksp_type chebyshev -%s_pc_type jacobi -%s_ksp_max_it %d -%s_ksp_norm_type none "
                 "-%s_ksp_chebyshev_eigenvalues %.17g,%.17g",
                 pre, pre, pre, coarse ? mg.opt.ncoarse : mg.opt.nsmooth, pre, pre, lo, hi);
        o += t;
    }
    if (buf && cap > 0) {
        const size_t n = o.size() < cap - 1 ? o.size() : cap - 1;
        memcpy(buf, o.data(), n);
        buf[n] = 0;
    }
    return (int)o.size();
}
// PCMGSetCycleType / PCMGSetCycleTypeOnLevel: cycles[l] cycles of level l + 1 per visit of level l (l = 0: the finest;
// 1 = V, 2 = W); entries beyond n keep their value.  Into the coarsest level there is always one (as in PETSc).
extern "C" int tp_elasticity_set_cycles(tp_elasticity *e, const int *cycles, int n) {
    if (!e || !cycles || n < 0 || n > TP_MAX_LEVELS) return TP_ERR_ARG;
    for (int l = 0; l < n; l++) {
        if (cycles[l] < 1 || cycles[l] > 4) return TP_ERR_ARG;
        e->mg.cycles[l] = cycles[l];
    }
    return TP_OK;
}
extern "C" int tp_elasticity_level_count(const tp_elasticity *e) { return e->mg.nlv; }
extern "C" long tp_elasticity_level_nodes(const tp_elasticity *e, int l) { return e->mg.lv[l].g.nodes(); }
extern "C" double tp_elasticity_level_lambda(const tp_elasticity *e, int l) { return e->mg.lv[l].lam; }
extern "C" double tp_elasticity_level_lambda_min(const tp_elasticity *e, int l) { return e->mg.lv[l].lam_min; }
extern "C" int tp_elasticity_coarse_direct_active(const tp_elasticity *e) { return e->assembled && e->mg.cd.factored ? e->mg.cd.g.n : 0; }
extern "C" int tp_elasticity_level_apply(tp_elasticity *e, int l, const double *u, double *y) {
    if (!e->assembled || l < 0 || l >= e->mg.nlv) return TP_ERR_STATE;
    return e->mg.apply(l, const_cast<double *>(u), y);
}
extern "C" int tp_elasticity_level_diag(tp_elasticity *e, int l, double *d) {
    if (!e->assembled) return TP_ERR_STATE;
    Level<3> &L = e->mg.lv[l];
    TP_HIP(hipMemcpyAsync(d, L.dinv, sizeof(double) * (size_t)L.ndof(), hipMemcpyDeviceToDevice, e->grid->stream));
    return TP_OK;
}
// ksp_mode 1 building blocks, level by level (tests): z = M^-1 r with PCJACOBI (pc 0) / PCSOR (pc 1), and the level's
// GMRES(m) run for its iterations on x (zero_guess: x is zeroed first; rtol < 0: no convergence test)
extern "C" int tp_elasticity_level_pc(tp_elasticity *e, int l, int pc, const double *r, double *z) {
    if (!e->assembled || l < 0 || l >= e->mg.nlv || e->mg.opt.ksp_mode != 1) return TP_ERR_STATE;
    RefKsp<3> *R;
    TP_TRY(refksp_get(e->mg, &R));
    return R->pc_apply(l, pc, r, z);
}
extern "C" int tp_elasticity_level_gmres(tp_elasticity *e, int l, int pc, int m, int its, double rtol, const double *b, double *x,
                                         int zero_guess, int *its_done) {
    if (!e->assembled || l < 0 || l >= e->mg.nlv || e->mg.opt.ksp_mode != 1) return TP_ERR_STATE;
    RefKsp<3> *R;
    TP_TRY(refksp_get(e->mg, &R));
    return R->gmres(l, b, x, zero_guess != 0, m, its, rtol < 0 ? 0.0 : rtol, e->mg.opt.atol, e->mg.opt.dtol, rtol >= 0, pc, its_done);
}
extern "C" int tp_elasticity_precond(tp_elasticity *e, const double *r, double *z) {
    if (!e->assembled) return TP_ERR_STATE;
    double *zp;
    TP_TRY(e->mg.precond(r, &zp));
    TP_HIP(hipMemcpyAsync(z, zp, sizeof(double) * (size_t)e->mg.lv[0].ndof(), hipMemcpyDeviceToDevice,
                          e->grid->stream));
    return TP_OK;
}
extern "C" int tp_elasticity_smooth(tp_elasticity *e, int l, const double *b, double *x, int k, int zero_guess) {
    if (!e->assembled || l < 0 || l >= e->mg.nlv) return TP_ERR_STATE;
    Level<3> &L = e->mg.lv[l];
    const size_t nb = sizeof(double) * (size_t)L.ndof();
    if (!zero_guess) TP_HIP(hipMemcpyAsync(L.x, x, nb, hipMemcpyDeviceToDevice, e->grid->stream));
    TP_TRY(e->mg.smooth(l, b, k, zero_guess != 0));
    TP_TRY(e->mg.drain_halos());
    TP_HIP(hipMemcpyAsync(x, L.x, nb, hipMemcpyDeviceToDevice, e->grid->stream));
    return TP_OK;
}
extern "C" int tp_elasticity_restrict(tp_elasticity *e, int l, const double *rf, double *rc) {
    MGSolver<3> &mg = e->mg;
    if (l < 0 || l + 1 >= mg.nlv) return TP_ERR_ARG;
    TP_TRY(mg.halo(l, const_cast<double *>(rf)));
    const Geom &C = mg.lv[l + 1].g;
    TP_LAUNCH((k_restrict<3>), dim3((int)((C.owned_nodes() + BLK - 1) / BLK)), dim3(BLK), 0,
                       e->grid->stream, C, mg.lv[l].g, rf, rc);
    return TP_OK;
}
extern "C" int tp_elasticity_prolong_add(tp_elasticity *e, int l, const double *xc, double *xf) {
    MGSolver<3> &mg = e->mg;
    if (l < 0 || l + 1 >= mg.nlv) return TP_ERR_ARG;
    TP_TRY(mg.halo(l + 1, const_cast<double *>(xc)));
    TP_LAUNCH((k_prolong_add<3>), dim3((int)((mg.lv[l].g.owned_nodes() + BLK - 1) / BLK)), dim3(BLK), 0,
                       e->grid->stream, mg.lv[l + 1].g, mg.lv[l].g, xc, xf);
    return TP_OK;
}
extern "C" int tp_elasticity_last_stats(const tp_elasticity *e, double *alg_bytes, double *flops, long *launches) {
    if (alg_bytes) *alg_bytes = e->grid->alg_bytes;
    if (flops) *flops = e->grid->flops;
    if (launches) *launches = e->grid->launches;
    e->grid->alg_bytes = e->grid->flops = 0.0;
    e->grid->launches = 0;
    return TP_OK;
}

// ===========================================================================
// density / sensitivity filter and Helmholtz PDE filter
// ===========================================================================
#include "filter.h"

// ===========================================================================
// optimizer step around the path (MMA), SURVEY.md 8(f)-1
// ===========================================================================
#include "mma.h"

// grid.h -- z-slab partition, halo exchange through the host framework's
// collectives, deterministic reductions with device-resident results.
#pragma once
#include <chrono>
#include <vector>
#include "common.h"

struct tp_grid {
    tp_grid_opts o;
    hipStream_t stream;
    tp_comm comm;
    bool has_comm;
    struct RcclComm *rccl = nullptr;  // set by tp_grid_use_rccl: the hooks above then point into it
    // halo overlap (DMGlobalToLocalBegin/End split, LinearElasticity.cc:249-250): ghost planes travel on a second
    // stream while the interior planes of the producing kernel are computed on `stream`
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_ready = nullptr;    // "boundary planes written" on `stream`
    bool overlap = false;
    long n_overlapped = 0;            // halos started with halo_nodes_begin
    tp_comm comm_host;                // the host framework's hooks given at creation (restored by tp_grid_drop_rccl)
    int ex, ey, ez_glob, ez_own;  // fine level element counts
    int rank, nranks;
    double *partials;   // [dev] MAX_RED_BLOCKS * 4
    double *scal;       // [dev] 64 device scalars
    unsigned *ticket;   // [dev] arrival counter of the in-kernel reduction tails (common.h: reduce_tail), rests at 0
    double *h_scal;     // pinned host mirror
    double *h_scal_dev; // its address as the device sees it (kernels that deposit a scalar for the host directly)
    hipEvent_t ev_scal; // marks the read-back of read_scal_begin
    // accounting (algorithmic model)
    double alg_bytes, flops;
    long launches;
    // opt-in timer of the roofline kernel IN PLACE (tp_grid_kernel_timer): a HIP event pair around every launch of the
    // fine level's fused Chebyshev step, on the stream it is launched on
    bool kt_on = false;
    double kt_bytes = 0.0;          // algorithmic bytes of the timed launches (per variant: with / without previous iterate)
    std::vector<hipEvent_t> kt_ev;  // pairs (start, stop)
    // opt-in timer of the communication (tp_grid_comm_timer; N > 1): per kind -- 0 blocking halo exchange on `stream`, 1 halo
    // exchange overlapped on `comm_stream`, 2 all-reduce, 3 all-gather of the replicated coarse levels -- the host wall time spent
    // inside the hook and a HIP event pair around it on the stream it is issued on (device time: what a host-staged hook hides
    // behind its own synchronisation shows in the first, what RCCL enqueues in the second)
    bool ct_on = false;
    std::vector<hipEvent_t> ct_ev[4];
    double ct_host_s[4] = {0, 0, 0, 0};
    long ct_calls[4] = {0, 0, 0, 0};
};
struct CommMark {
    tp_grid *g;
    int kind;
    hipStream_t s;
    std::chrono::steady_clock::time_point t0;
    CommMark(tp_grid *g_, int kind_, hipStream_t s_) : g(g_), kind(kind_), s(s_) {
        if (!g->ct_on) return;
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) == hipSuccess) {
            (void)hipEventRecord(e, s);
            g->ct_ev[kind].push_back(e);
        }
        t0 = std::chrono::steady_clock::now();
    }
    ~CommMark() {
        if (!g->ct_on) return;
        g->ct_host_s[kind] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        g->ct_calls[kind]++;
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) == hipSuccess) {
            (void)hipEventRecord(e, s);
            g->ct_ev[kind].push_back(e);
        }
    }
};
inline void kernel_timer_mark(tp_grid *g) {
    if (!g->kt_on) return;
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, g->stream);
    g->kt_ev.push_back(e);
}

// geometry of multigrid level `l` of this rank's slab
inline Geom make_geom(const tp_grid *g, int l) {
    Geom q;
    q.ex = g->ex >> l;
    q.ey = g->ey >> l;
    q.nx = q.ex + 1;
    q.ny = q.ey + 1;
    q.ez_own = g->ez_own >> l;
    q.has_lo = g->rank > 0;
    q.has_hi = g->rank < g->nranks - 1;
    q.ezl = q.ez_own + q.has_hi;
    q.nzl = q.ez_own + 1 + q.has_hi;
    q.own_lo = q.has_lo ? 1 : 0;
    q.own_hi = q.ez_own;
    q.gz0 = g->rank * q.ez_own;
    q.nz_glob = (g->ez_glob >> l) + 1;
    return q;
}

inline void count_launch(tp_grid *g, double bytes = 0.0, double flops = 0.0) {
    g->launches++;
    g->alg_bytes += bytes;
    g->flops += flops;
}

// scal[slot .. slot+NV) holds this rank's sums (written by the producing kernel's reduce_tail): sum over ranks
template <int NV>
inline int finish_reduction(tp_grid *g, int slot) {
    if (g->has_comm) {
        CommMark cm(g, 2, g->stream);
        if (g->comm.allreduce_inplace) {  // the CG scalars are reduced where they live
            if (g->comm.allreduce_inplace(g->comm.user, g->scal + slot, NV)) return TP_ERR_COMM;
        } else {
            TP_HIP(hipMemcpyAsync(g->comm.red, g->scal + slot, sizeof(double) * NV, hipMemcpyDeviceToDevice, g->stream));
            if (g->comm.allreduce_sum(g->comm.user, NV)) return TP_ERR_COMM;
            TP_HIP(hipMemcpyAsync(g->scal + slot, g->comm.red, sizeof(double) * NV, hipMemcpyDeviceToDevice, g->stream));
        }
    }
    return TP_OK;
}

// The reductions finished inside the producing kernel (reduce_tail, common.h) rely on relaxed agent-scope atomics and an
// explicit vmcnt(0) instead of a release fence (measured: a fence per workgroup writes back the whole L2).  Should that
// ever misbehave on another driver or firmware, TP_NO_REDUCE_TAIL=1 routes every such reduction through the former second
// launch (k_reduce_final: same summation order, same bits): the kernels get no ticket and leave their partial sums only.
inline unsigned *tail_ticket(tp_grid *g) {
    static const bool off = getenv("TP_NO_REDUCE_TAIL") != nullptr;
    return off ? nullptr : g->ticket;
}
template <int NV>
inline int finish_reduction(tp_grid *g, int slot);
// behind a kernel that ends in reduce_tail<NV> with nblocks workgroups: second launch if the tail is switched off, ranks
template <int NV>
inline int finish_tail(tp_grid *g, int nblocks, int slot) {
    if (!tail_ticket(g)) {
        TP_LAUNCH(k_reduce_final<NV>, dim3(1), dim3(BLK), 0, g->stream, g->partials, nblocks, g->scal + slot);
        count_launch(g);
    }
    return finish_reduction<NV>(g, slot);
}

// the two-launch form for the once-per-call reductions (block partials in g->partials -> scal[slot..], then ranks)
template <int NV>
inline int reduce_partials(tp_grid *g, int nblocks, int slot) {
    TP_LAUNCH(k_reduce_final<NV>, dim3(1), dim3(BLK), 0, g->stream, g->partials, nblocks, g->scal + slot);
    count_launch(g);
    return finish_reduction<NV>(g, slot);
}

// blocking read of device scalars (the only host synchronisation of the Krylov loop)
inline int read_scal(tp_grid *g, int slot, int n, double *out) {
    TP_HIP(hipMemcpyAsync(g->h_scal, g->scal + slot, sizeof(double) * n, hipMemcpyDeviceToHost, g->stream));
    TP_HIP(hipStreamSynchronize(g->stream));
    for (int i = 0; i < n; i++) out[i] = g->h_scal[i];
    return TP_OK;
}

// the same in two halves: the copy is enqueued (and marked by an event) where the value is ready, more work may be
// enqueued behind it, and the host waits for the event only -- not for what it put on the stream in between
inline int read_scal_begin(tp_grid *g, int slot, int n) {
    TP_HIP(hipMemcpyAsync(g->h_scal, g->scal + slot, sizeof(double) * n, hipMemcpyDeviceToHost, g->stream));
    if (!g->ev_scal) TP_HIP(hipEventCreateWithFlags(&g->ev_scal, hipEventDisableTiming));
    TP_HIP(hipEventRecord(g->ev_scal, g->stream));
    return TP_OK;
}
inline int read_scal_end(tp_grid *g, int n, double *out) {
    TP_HIP(hipEventSynchronize(g->ev_scal));
    for (int i = 0; i < n; i++) out[i] = g->h_scal[i];
    return TP_OK;
}

inline int dot_to_slot(tp_grid *g, const double *a, const double *b, long n, int slot) {
    int nb = grid_for(n, 2048);
    TP_LAUNCH(k_dot, dim3(nb), dim3(BLK), 0, g->stream, a, b, n, g->partials, tail_ticket(g), g->scal + slot);
    count_launch(g, 16.0 * n, 2.0 * n);
    return finish_tail<1>(g, nb, slot);
}
inline int sum_to_slot(tp_grid *g, const double *a, long n, int slot) {
    int nb = grid_for(n, 2048);
    TP_LAUNCH(k_sum, dim3(nb), dim3(BLK), 0, g->stream, a, n, g->partials, tail_ticket(g), g->scal + slot);
    count_launch(g, 8.0 * n, 1.0 * n);
    return finish_tail<1>(g, nb, slot);
}

// Generic neighbour exchange of `rows` segments of `seg` doubles each (pitch in
// doubles), through the framework-owned staging buffers:
//   to_lo   -> rank-1 (arrives there as from_hi),  to_hi -> rank+1 (arrives as from_lo).
// Any of the four pointers may be NULL.  Stream ordered.
inline int exchange_segments(tp_grid *g, const double *to_lo, double *from_lo, const double *to_hi, double *from_hi,
                             long seg, long rows, long pitch) {
    if (!g->has_comm || seg <= 0 || rows <= 0) return TP_OK;
    const tp_comm &c = g->comm;
    long rows_per = c.cap / seg;
    if (rows_per < 1) return TP_ERR_ARG;
    for (long r0 = 0; r0 < rows; r0 += rows_per) {
        long nr = rows - r0 < rows_per ? rows - r0 : rows_per;
        if (to_lo && g->rank > 0)
            TP_HIP(hipMemcpy2DAsync(c.send_lo, seg * 8, to_lo + r0 * pitch, pitch * 8, seg * 8, nr,
                                    hipMemcpyDeviceToDevice, g->stream));
        if (to_hi && g->rank < g->nranks - 1)
            TP_HIP(hipMemcpy2DAsync(c.send_hi, seg * 8, to_hi + r0 * pitch, pitch * 8, seg * 8, nr,
                                    hipMemcpyDeviceToDevice, g->stream));
        {
            CommMark cm(g, 0, g->stream);
            if (c.exchange(c.user, seg * nr)) return TP_ERR_COMM;
        }
        if (from_lo && g->rank > 0)
            TP_HIP(hipMemcpy2DAsync(from_lo + r0 * pitch, pitch * 8, c.recv_lo, seg * 8, seg * 8, nr,
                                    hipMemcpyDeviceToDevice, g->stream));
        if (from_hi && g->rank < g->nranks - 1)
            TP_HIP(hipMemcpy2DAsync(from_hi + r0 * pitch, pitch * 8, c.recv_hi, seg * 8, seg * 8, nr,
                                    hipMemcpyDeviceToDevice, g->stream));
    }
    return TP_OK;
}

inline bool halo_can_overlap(const tp_grid *g) {
    return g->has_comm && g->overlap && g->comm_stream && g->comm.set_stream && g->comm.exchange_direct;
}
// DMGlobalToLocalBegin: the boundary planes of v are complete on g->stream at the time of the call; the ghost planes
// are exchanged (zero-copy, in place) on the second stream and `done` is recorded there -- the consumer of the ghost
// planes makes g->stream wait for it (MGSolver::halo).  Returns 2 if the host cannot exchange in place (nothing was
// sent: the caller falls back to the blocking form for good).
inline int halo_nodes_begin(tp_grid *g, const Geom &q, double *v, int dof, hipEvent_t done) {
    const long pl = q.plane() * dof;
    const bool lo = g->rank > 0, hi = g->rank < g->nranks - 1;
    TP_HIP(hipEventRecord(g->ev_ready, g->stream));
    TP_HIP(hipStreamWaitEvent(g->comm_stream, g->ev_ready, 0));
    g->comm.set_stream(g->comm.user, g->comm_stream);
    int rc;
    {
        CommMark cm(g, 1, g->comm_stream);
        rc = g->comm.exchange_direct(g->comm.user, lo ? v + pl * q.own_lo : nullptr, lo ? v : nullptr,
                                     hi ? v + pl * q.own_hi : nullptr, hi ? v + pl * (q.nzl - 1) : nullptr, pl);
    }
    g->comm.set_stream(g->comm.user, nullptr);
    if (rc == 2) {
        g->overlap = false;
        g->comm.exchange_direct = nullptr;
        return 2;
    }
    if (rc) return TP_ERR_COMM;
    TP_HIP(hipEventRecord(done, g->comm_stream));
    g->n_overlapped++;
    return TP_OK;
}

// refresh the ghost node planes of a level vector (DMGlobalToLocalBegin/End,
// LinearElasticity.cc:249-250): own bottom plane -> lower neighbour's top ghost,
// own top plane -> upper neighbour's bottom ghost.
inline int halo_nodes(tp_grid *g, const Geom &q, double *v, int dof) {
    if (!g->has_comm) return TP_OK;
    long pl = q.plane() * dof;
    if (g->comm.exchange_direct) {  // zero-copy: the framework sends/receives the planes in place
        const bool lo = g->rank > 0, hi = g->rank < g->nranks - 1;
        int rc;
        {
            CommMark cm(g, 0, g->stream);
            rc = g->comm.exchange_direct(g->comm.user, lo ? v + pl * q.own_lo : nullptr, lo ? v : nullptr,
                                         hi ? v + pl * q.own_hi : nullptr, hi ? v + pl * (q.nzl - 1) : nullptr, pl);
        }
        if (rc == 0) return TP_OK;
        if (rc != 2) return TP_ERR_COMM;
        g->comm.exchange_direct = nullptr;  // 2: the host cannot address our memory in place -> staging from now on
    }
    return exchange_segments(g, v + pl * q.own_lo, v /*plane 0*/, v + pl * q.own_hi, v + pl * (q.nzl - 1), pl, 1, pl);
}

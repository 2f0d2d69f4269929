// rccl_comm.h -- the slab exchange of tp_comm issued DIRECTLY to RCCL on the solver's stream.
//
// The default hooks (include/topopt_amd.h: tp_comm) call back into the host framework (torch.distributed),
// which costs a Python round trip per halo.  On one node the neighbour exchange is a grouped ncclSend/ncclRecv
// pair per face over xGMI and the reductions are ncclAllReduce/ncclAllGather: a few microseconds of host time
// each when issued from here.  RCCL is not linked: the framework passes the path of the librccl.so it already
// uses (one RCCL instance per process) and the entry points are resolved with dlsym.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "grid.h"

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;  // optional: what the communicator itself says it spans
};
inline RcclApi &rccl_api() {
    static RcclApi a;
    return a;
}

struct RcclComm {
    ncclComm_t comm = nullptr;          // collectives (all-reduce, all-gather) of the solver's stream
    // Neighbour exchanges get a communicator of their own: RCCL orders the operations of ONE communicator, so a halo
    // in flight on the second stream and an all-reduce issued on the solver's stream would run one behind the other
    // and the overlap of the halo with the interior planes (grid.h) would not survive contact with hardware.
    // Null (tp_grid_use_rccl with one id): both kinds share `comm` as in round 2.
    ncclComm_t comm_halo = nullptr;
    hipStream_t stream = nullptr;       // stream the operations are issued on (set_stream hook)
    hipStream_t home_stream = nullptr;  // the grid's stream
    int rank = 0, nranks = 1;
    bool periodic = false;  // self test only: the single rank is its own lower and upper neighbour
    double *buf = nullptr;  // owns send_lo | send_hi | recv_lo | recv_hi | red | gather
    tp_comm hooks{};
    long n_exchanges = 0, n_reductions = 0;
};

#define TP_NCCL(call)                                                                                         \
    do {                                                                                                      \
        ncclResult_t _r = (call);                                                                             \
        if (_r != ncclSuccess) {                                                                              \
            fprintf(stderr, "topopt_amd: %s -> %s\n", #call, rccl_api().GetErrorString ? rccl_api().GetErrorString(_r) : "?"); \
            return 1;                                                                                         \
        }                                                                                                     \
    } while (0)

static int rccl_pairs(RcclComm *c, const double *to_lo, double *from_lo, const double *to_hi, double *from_hi, long n) {
    RcclApi &A = rccl_api();
    const int lo = c->periodic ? c->rank : c->rank - 1, hi = c->periodic ? c->rank : c->rank + 1;
    const bool has_lo = c->periodic || lo >= 0, has_hi = c->periodic || hi < c->nranks;
    ncclComm_t cm = c->comm_halo ? c->comm_halo : c->comm;
    TP_NCCL(A.GroupStart());
    if (has_hi && to_hi) TP_NCCL(A.Send(to_hi, (size_t)n, ncclDouble, hi, cm, c->stream));
    if (has_lo && from_lo) TP_NCCL(A.Recv(from_lo, (size_t)n, ncclDouble, lo, cm, c->stream));
    if (has_lo && to_lo) TP_NCCL(A.Send(to_lo, (size_t)n, ncclDouble, lo, cm, c->stream));
    if (has_hi && from_hi) TP_NCCL(A.Recv(from_hi, (size_t)n, ncclDouble, hi, cm, c->stream));
    TP_NCCL(A.GroupEnd());
    c->n_exchanges++;
    return 0;
}
static int rccl_exchange(void *u, long n) {
    RcclComm *c = (RcclComm *)u;
    return rccl_pairs(c, c->hooks.send_lo, c->hooks.recv_lo, c->hooks.send_hi, c->hooks.recv_hi, n);
}
static int rccl_exchange_direct(void *u, const double *to_lo, double *from_lo, const double *to_hi, double *from_hi,
                                long n) {
    return rccl_pairs((RcclComm *)u, to_lo, from_lo, to_hi, from_hi, n);
}
static int rccl_allreduce(void *u, int n) {
    RcclComm *c = (RcclComm *)u;
    TP_NCCL(rccl_api().AllReduce(c->hooks.red, c->hooks.red, (size_t)n, ncclDouble, ncclSum, c->comm, c->stream));
    c->n_reductions++;
    return 0;
}
static int rccl_allreduce_inplace(void *u, double *p, int n) {
    RcclComm *c = (RcclComm *)u;
    TP_NCCL(rccl_api().AllReduce(p, p, (size_t)n, ncclDouble, ncclSum, c->comm, c->stream));
    c->n_reductions++;
    return 0;
}
static void rccl_set_stream(void *u, void *stream) {
    RcclComm *c = (RcclComm *)u;
    c->stream = stream ? (hipStream_t)stream : c->home_stream;
}
static int rccl_allgather(void *u, long n) {
    RcclComm *c = (RcclComm *)u;
    TP_NCCL(rccl_api().AllGather(c->hooks.send_lo, c->hooks.gather, (size_t)n, ncclDouble, c->comm, c->stream));
    return 0;
}

inline int rccl_load(const char *path) {
    RcclApi &A = rccl_api();
    if (A.handle) return TP_OK;
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        fprintf(stderr, "topopt_amd: dlopen(%s): %s\n", path ? path : "(null)", dlerror());
        return TP_ERR_COMM;
    }
#define TP_SYM(field, name)                                     \
    *(void **)(&A.field) = dlsym(h, name);                      \
    if (!A.field) {                                             \
        fprintf(stderr, "topopt_amd: %s not found in %s\n", name, path); \
        dlclose(h);                                             \
        A = RcclApi();                                          \
        return TP_ERR_COMM;                                     \
    }
    TP_SYM(GetUniqueId, "ncclGetUniqueId")
    TP_SYM(CommInitRank, "ncclCommInitRank")
    TP_SYM(CommDestroy, "ncclCommDestroy")
    TP_SYM(GroupStart, "ncclGroupStart")
    TP_SYM(GroupEnd, "ncclGroupEnd")
    TP_SYM(Send, "ncclSend")
    TP_SYM(Recv, "ncclRecv")
    TP_SYM(AllReduce, "ncclAllReduce")
    TP_SYM(AllGather, "ncclAllGather")
    TP_SYM(GetErrorString, "ncclGetErrorString")
#undef TP_SYM
    *(void **)(&A.CommCount) = dlsym(h, "ncclCommCount");  // (reporting only)
    A.handle = h;
    return TP_OK;
}

// collective over the ranks that hold the same unique id; cap = doubles per staging buffer
// id128_halo (may be null): a second unique id for the communicator of the neighbour exchanges
inline int rccl_comm_create(RcclComm **out, const void *id128, int rank, int nranks, int device, hipStream_t stream,
                            long cap, const void *id128_halo = nullptr) {
    RcclApi &A = rccl_api();
    if (!A.handle || !out || !id128 || cap < 16) return TP_ERR_ARG;
    TP_HIP(hipSetDevice(device));
    RcclComm *c = new RcclComm();
    c->rank = rank;
    c->nranks = nranks;
    c->stream = c->home_stream = stream;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    if (A.CommInitRank(&c->comm, nranks, id, rank) != ncclSuccess) {
        delete c;
        return TP_ERR_COMM;
    }
    if (id128_halo) {
        ncclUniqueId id2;
        memcpy(&id2, id128_halo, sizeof(id2));
        if (A.CommInitRank(&c->comm_halo, nranks, id2, rank) != ncclSuccess) {
            A.CommDestroy(c->comm);
            delete c;
            return TP_ERR_COMM;
        }
    }
    const size_t total = (size_t)cap * (4 + (size_t)nranks) + 16;
    if (hipMalloc((void **)&c->buf, sizeof(double) * total) != hipSuccess) {
        if (c->comm_halo) A.CommDestroy(c->comm_halo);
        A.CommDestroy(c->comm);
        delete c;
        return TP_ERR_HIP + (int)hipErrorOutOfMemory;
    }
    (void)hipMemsetAsync(c->buf, 0, sizeof(double) * total, stream);
    tp_comm &h = c->hooks;
    h.user = c;
    h.send_lo = c->buf;
    h.send_hi = c->buf + cap;
    h.recv_lo = c->buf + 2 * cap;
    h.recv_hi = c->buf + 3 * cap;
    h.gather = c->buf + 4 * cap;
    h.red = c->buf + (4 + (size_t)nranks) * cap;
    h.cap = cap;
    h.exchange = rccl_exchange;
    h.allreduce_sum = rccl_allreduce;
    h.allgather = rccl_allgather;
    h.exchange_direct = rccl_exchange_direct;
    h.allreduce_inplace = rccl_allreduce_inplace;
    h.set_stream = rccl_set_stream;
    *out = c;
    return TP_OK;
}
inline void rccl_comm_destroy(RcclComm *c) {
    if (!c) return;
    (void)hipStreamSynchronize(c->home_stream);
    if (c->comm_halo) rccl_api().CommDestroy(c->comm_halo);
    if (c->comm) rccl_api().CommDestroy(c->comm);
    (void)hipFree(c->buf);
    delete c;
}

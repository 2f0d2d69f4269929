// slab_comm.h -- tp_comm hooks for a C++ host with one process per GPU on ONE node, without MPI and without torch:
//   * a launcher (host/slabrun) starts N copies of the program with TP_RANK / TP_NRANKS / TP_SHM / TP_DEVICE set;
//   * the processes meet in a POSIX shared-memory segment (a barrier and one mailbox per rank);
//   * the hooks stage through the host (hipMemcpy to the mailbox, barrier, hipMemcpy from the neighbours'): correct
//     everywhere -- including N ranks on one GPU, which RCCL refuses -- and slow;
//   * slab_comm_try_rccl() then hands the exchange to RCCL inside the library (tp_grid_use_rccl, topopt_amd.h): rank 0
//     makes the unique id, the mailboxes carry it to the others, the library checks the path collectively and falls
//     back to these hooks if any rank cannot use it.
// It is also the reduction layer of the multi-process PETSc-named surface (host/shim/sys.cc: MPI_Allreduce & co.).
// What an MPI host would write instead: the same three hooks with MPI_Sendrecv / MPI_Allreduce / MPI_Allgather on
// host-staged or GPU-aware buffers.
#ifndef TOPOPT_SLAB_COMM_H
#define TOPOPT_SLAB_COMM_H
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "topopt_amd.h"

struct SlabShmHeader {
    std::atomic<int> arrive;
    std::atomic<int> gen;
    std::atomic<int> attached;
    int nranks;
    long slot_doubles;
};

struct SlabComm {
    int rank = 0, nranks = 1, device = 0;
    SlabShmHeader *hdr = nullptr;
    double *slots = nullptr;  // nranks x (2 * slot_doubles): [lo | hi] mailbox of every rank
    size_t map_bytes = 0;
    std::string shm_name;
    tp_comm hooks{};
    double *dev_buf = nullptr;  // send_lo | send_hi | recv_lo | recv_hi | red(16) | gather(nranks * cap)
    tp_grid *grid = nullptr;    // set by the host once the grid exists (stream synchronisation)
    long n_exchanges = 0;

    double *mailbox(int r, int half) const { return slots + ((size_t)r * 2 + half) * (size_t)hdr->slot_doubles; }
    // all ranks of the job; gives up (and kills the job) after 120 s instead of hanging the GPU box
    void barrier() {
        if (nranks == 1) return;
        const int g = hdr->gen.load(std::memory_order_acquire);
        if (hdr->arrive.fetch_add(1, std::memory_order_acq_rel) + 1 == nranks) {
            hdr->arrive.store(0, std::memory_order_relaxed);
            hdr->gen.fetch_add(1, std::memory_order_acq_rel);
            return;
        }
        const auto t0 = std::chrono::steady_clock::now();
        long spins = 0;
        while (hdr->gen.load(std::memory_order_acquire) == g) {
            if ((++spins & 1023) == 0) {
                sched_yield();
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) {
                    fprintf(stderr, "slab_comm: rank %d waited 120 s in a barrier -- a peer is gone; aborting\n", rank);
                    _exit(86);
                }
            }
        }
    }
};

namespace slab_detail {
inline int sync_stream(SlabComm *c) { return c->grid ? tp_sync(c->grid) : 0; }
inline int hook_exchange(void *u, long n) {
    SlabComm *c = (SlabComm *)u;
    if (n > c->hooks.cap || sync_stream(c)) return 1;
    if (c->rank > 0 && tp_memcpy_d2h(c->mailbox(c->rank, 0), c->hooks.send_lo, sizeof(double) * (size_t)n)) return 1;
    if (c->rank < c->nranks - 1 && tp_memcpy_d2h(c->mailbox(c->rank, 1), c->hooks.send_hi, sizeof(double) * (size_t)n)) return 1;
    c->barrier();
    if (c->rank > 0 && tp_memcpy_h2d(c->hooks.recv_lo, c->mailbox(c->rank - 1, 1), sizeof(double) * (size_t)n)) return 1;
    if (c->rank < c->nranks - 1 && tp_memcpy_h2d(c->hooks.recv_hi, c->mailbox(c->rank + 1, 0), sizeof(double) * (size_t)n)) return 1;
    c->barrier();
    c->n_exchanges++;
    return 0;
}
// sum / max / min over ranks of n host doubles, in rank order on every rank (bitwise the same everywhere)
inline void host_reduce(SlabComm *c, double *v, int n, int op /*0 sum, 1 max, 2 min*/) {
    if (c->nranks == 1) return;
    memcpy(c->mailbox(c->rank, 0), v, sizeof(double) * (size_t)n);
    c->barrier();
    for (int i = 0; i < n; i++) {
        double s = c->mailbox(0, 0)[i];
        for (int r = 1; r < c->nranks; r++) {
            const double w = c->mailbox(r, 0)[i];
            s = op == 0 ? s + w : (op == 1 ? (w > s ? w : s) : (w < s ? w : s));
        }
        v[i] = s;
    }
    c->barrier();
}
inline int hook_allreduce(void *u, int n) {
    SlabComm *c = (SlabComm *)u;
    double v[16];
    if (n > 16 || sync_stream(c) || tp_memcpy_d2h(v, c->hooks.red, sizeof(double) * (size_t)n)) return 1;
    host_reduce(c, v, n, 0);
    return tp_memcpy_h2d(c->hooks.red, v, sizeof(double) * (size_t)n);
}
inline int hook_allgather(void *u, long n) {
    SlabComm *c = (SlabComm *)u;
    if (n > c->hooks.cap || sync_stream(c)) return 1;
    if (tp_memcpy_d2h(c->mailbox(c->rank, 0), c->hooks.send_lo, sizeof(double) * (size_t)n)) return 1;
    c->barrier();
    for (int r = 0; r < c->nranks; r++)
        if (tp_memcpy_h2d(c->hooks.gather + (size_t)r * (size_t)n, c->mailbox(r, 0), sizeof(double) * (size_t)n)) return 1;
    c->barrier();
    return 0;
}
}  // namespace slab_detail

// Joins the job described by TP_RANK / TP_NRANKS / TP_SHM (absent: a one-rank job): mailboxes of `cap` doubles per
// half, no device memory yet (this much runs without a GPU: host/slab_selftest.cc).  Returns 0, or an error code after a
// message on stderr.
inline int slab_comm_join(SlabComm *c, long cap) {
    const char *er = getenv("TP_RANK"), *en = getenv("TP_NRANKS"), *es = getenv("TP_SHM"), *ed = getenv("TP_DEVICE");
    c->rank = er ? atoi(er) : 0;
    c->nranks = en ? atoi(en) : 1;
    c->device = ed ? atoi(ed) : 0;
    if (c->nranks < 1 || c->rank < 0 || c->rank >= c->nranks) {
        fprintf(stderr, "slab_comm: bad TP_RANK / TP_NRANKS\n");
        return 1;
    }
    if (cap < 16) cap = 16;
    if (c->nranks > 1) {
        if (!es) {
            fprintf(stderr, "slab_comm: TP_NRANKS > 1 needs TP_SHM (start the program with host/slabrun)\n");
            return 1;
        }
        c->shm_name = es;
        c->map_bytes = sizeof(SlabShmHeader) + 64 + sizeof(double) * 2 * (size_t)cap * (size_t)c->nranks;
        int fd = -1;
        if (c->rank == 0) {
            fd = shm_open(es, O_CREAT | O_RDWR, 0600);
            if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes)) {
                perror("slab_comm: shm_open");
                return 1;
            }
        } else {
            for (int t = 0; t < 60000 && fd < 0; t++) {  // wait for rank 0 (60 s)
                fd = shm_open(es, O_RDWR, 0600);
                struct stat st;
                if (fd >= 0 && (fstat(fd, &st) || (size_t)st.st_size < c->map_bytes)) {
                    close(fd);
                    fd = -1;
                }
                if (fd < 0) usleep(1000);
            }
            if (fd < 0) {
                fprintf(stderr, "slab_comm: rank %d: no segment %s\n", c->rank, es);
                return 1;
            }
        }
        void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) {
            perror("slab_comm: mmap");
            return 1;
        }
        c->hdr = (SlabShmHeader *)p;
        c->slots = (double *)((char *)p + ((sizeof(SlabShmHeader) + 63) / 64) * 64);
        if (c->rank == 0) {  // a fresh segment is zero-filled: counters start at 0
            c->hdr->nranks = c->nranks;
            c->hdr->slot_doubles = cap;
        }
        c->hdr->attached.fetch_add(1);
        for (int t = 0; c->hdr->attached.load() < c->nranks; t++) {
            if (t > 60000) {
                fprintf(stderr, "slab_comm: rank %d: only %d of %d ranks attached\n", c->rank, c->hdr->attached.load(), c->nranks);
                return 1;
            }
            usleep(1000);
        }
        if (c->rank == 0) shm_unlink(es);  // everybody is mapped: the name can go
    } else {
        static SlabShmHeader one;
        one.nranks = 1;
        one.slot_doubles = cap;
        c->hdr = &one;
    }
    c->hooks.cap = cap;
    return 0;
}
// the device staging buffers of the tp_comm hooks on the rank's GPU (after slab_comm_join)
inline int slab_comm_alloc(SlabComm *c) {
    if (c->dev_buf) return 0;
    const long cap = c->hooks.cap;
    const size_t nd = 4 * (size_t)cap + 16 + (size_t)c->nranks * (size_t)cap;
    if (tp_set_device(c->device) || tp_malloc((void **)&c->dev_buf, sizeof(double) * nd)) return 1;
    tp_comm &h = c->hooks;
    h.user = c;
    h.send_lo = c->dev_buf;
    h.send_hi = h.send_lo + cap;
    h.recv_lo = h.send_hi + cap;
    h.recv_hi = h.recv_lo + cap;
    h.red = h.recv_hi + cap;
    h.gather = h.red + 16;
    h.cap = cap;
    h.exchange = slab_detail::hook_exchange;
    h.allreduce_sum = slab_detail::hook_allreduce;
    h.allgather = slab_detail::hook_allgather;
    h.exchange_direct = nullptr;
    h.allreduce_inplace = nullptr;
    h.set_stream = nullptr;
    return 0;
}
inline int slab_comm_init(SlabComm *c, long cap) { return slab_comm_join(c, cap) || slab_comm_alloc(c); }

// Optional upgrade to the library's own RCCL path (one process per GPU on real hardware).  Harmless when it cannot be
// used (ranks sharing a GPU, no librccl): every rank keeps the host-staged hooks.  TP_RCCL_LIB names the library.
inline void slab_comm_try_rccl(SlabComm *c, tp_grid *g) {
    if (c->nranks == 1 || getenv("TP_NO_RCCL")) return;
    const char *path = getenv("TP_RCCL_LIB") ? getenv("TP_RCCL_LIB") : "/opt/rocm/lib/librccl.so";
    double ok = tp_rccl_load(path) == 0 ? 1.0 : 0.0;
    slab_detail::host_reduce(c, &ok, 1, 2);
    if (ok == 0.0) return;
    char id[128];
    memset(id, 0, sizeof(id));
    double got = 1.0;
    if (c->rank == 0) {
        got = tp_rccl_unique_id(id) == 0 ? 1.0 : 0.0;
        memcpy(c->mailbox(0, 1), id, sizeof(id));
    }
    slab_detail::host_reduce(c, &got, 1, 2);  // (its barriers also publish the id)
    if (got == 0.0) return;
    memcpy(id, c->mailbox(0, 1), sizeof(id));
    c->barrier();
    // every rank takes the same decision: the communicator must exist everywhere, and rank-tagged planes through the new
    // path must arrive where they belong -- else all ranks return to the mailboxes
    double good = tp_grid_use_rccl(g, id) == 0 ? 1.0 : 0.0;
    slab_detail::host_reduce(c, &good, 1, 2);
    if (good != 0.0) {
        int ok = 0;
        good = (tp_grid_comm_selfcheck(g, &ok) == 0 && ok == 1) ? 1.0 : 0.0;
        slab_detail::host_reduce(c, &good, 1, 2);
    }
    if (good == 0.0) (void)tp_grid_drop_rccl(g);
}

inline void slab_comm_free(SlabComm *c) {
    if (c->dev_buf) tp_free(c->dev_buf);
    c->dev_buf = nullptr;
    if (c->nranks > 1 && c->hdr) munmap((void *)c->hdr, c->map_bytes);
    c->hdr = nullptr;
}
#endif

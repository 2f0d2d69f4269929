// slab_selftest -- the host side of slab_comm.h without a GPU: joins the job that slabrun started, pushes rank-tagged
// data through the barrier, the rank-ordered reductions and the mailboxes (the neighbour pattern of the halo hook), and
// prints "rank r of n OK".  tests/test_cpp_host.py runs it with 1, 2, 3 and 5 ranks on the CPU.
#include <cmath>
#include <cstdio>

#include "slab_comm.h"

// the library calls of slab_comm.h that this program never reaches (no device buffers are allocated)
extern "C" {
int tp_sync(const tp_grid *) { return 0; }
int tp_memcpy_d2h(void *, const void *, size_t) { return 1; }
int tp_memcpy_h2d(void *, const void *, size_t) { return 1; }
int tp_set_device(int) { return 1; }
int tp_malloc(void **, size_t) { return 1; }
int tp_free(void *) { return 0; }
int tp_rccl_load(const char *) { return 1; }
int tp_rccl_unique_id(void *) { return 1; }
int tp_grid_use_rccl(tp_grid *, const void *) { return 1; }
int tp_grid_drop_rccl(tp_grid *) { return 0; }
int tp_grid_comm_selfcheck(tp_grid *, int *) { return 1; }
}

int main() {
    SlabComm c;
    if (slab_comm_join(&c, 64)) return 1;
    const int r = c.rank, n = c.nranks;
    int bad = 0;
    for (int round = 0; round < 200; round++) {
        // reductions: sum, max, min of rank-dependent values, identical on every rank
        double v[3] = {0.1 * (r + 1) + round, (double)((r * 7 + round) % n), (double)((r * 5 + round) % n)};
        double s = 0, mx = -1, mn = 1e9;
        for (int q = 0; q < n; q++) {
            s += 0.1 * (q + 1) + round;
            mx = std::fmax(mx, (double)((q * 7 + round) % n));
            mn = std::fmin(mn, (double)((q * 5 + round) % n));
        }
        slab_detail::host_reduce(&c, &v[0], 1, 0);
        slab_detail::host_reduce(&c, &v[1], 1, 1);
        slab_detail::host_reduce(&c, &v[2], 1, 2);
        if (v[0] != s || v[1] != mx || v[2] != mn) bad++;
        // neighbour exchange through the mailboxes, as hook_exchange does it (lo half -> rank-1, hi half -> rank+1)
        if (n > 1) {
            for (int i = 0; i < 8; i++) {
                c.mailbox(r, 0)[i] = 1000.0 * r + round + i;        // for the lower neighbour
                c.mailbox(r, 1)[i] = 1000.0 * r + round + i + 0.5;  // for the upper neighbour
            }
            c.barrier();
            for (int i = 0; i < 8; i++) {
                if (r > 0 && c.mailbox(r - 1, 1)[i] != 1000.0 * (r - 1) + round + i + 0.5) bad++;
                if (r < n - 1 && c.mailbox(r + 1, 0)[i] != 1000.0 * (r + 1) + round + i) bad++;
            }
            c.barrier();
        }
    }
    printf("rank %d of %d %s\n", r, n, bad ? "FAILED" : "OK");
    slab_comm_free(&c);
    return bad ? 1 : 0;
}

// slabrun -- starts N copies of a program as the ranks of one z-slab job on this node (host/slab_comm.h):
//   slabrun -n N [--same-device] program [args...]
// Every copy gets TP_RANK, TP_NRANKS, TP_SHM (a fresh shared-memory name) and TP_DEVICE (= rank, or 0 with
// --same-device: N slabs on one GPU, the way the one-GPU tests exercise the multi-rank path).  Rank 0's exit code is
// returned if all ranks succeed, otherwise the first failure; a rank that dies takes the others down (they time out
// in their next barrier, or are killed here).
#include <signal.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int main(int argc, char **argv) {
    int n = 1, a = 1;
    bool same = false;
    while (a < argc && argv[a][0] == '-') {
        if (!strcmp(argv[a], "-n") && a + 1 < argc) {
            n = atoi(argv[a + 1]);
            a += 2;
        } else if (!strcmp(argv[a], "--same-device")) {
            same = true;
            a++;
        } else {
            break;
        }
    }
    if (a >= argc || n < 1) {
        fprintf(stderr, "usage: slabrun -n N [--same-device] program [args...]\n");
        return 2;
    }
    const std::string shm = "/tp_slab_" + std::to_string((long)getpid());
    std::vector<pid_t> kids;
    for (int r = 0; r < n; r++) {
        const pid_t p = fork();
        if (p < 0) {
            perror("fork");
            for (pid_t k : kids) kill(k, SIGTERM);
            return 2;
        }
        if (p == 0) {
            setenv("TP_RANK", std::to_string(r).c_str(), 1);
            setenv("TP_NRANKS", std::to_string(n).c_str(), 1);
            setenv("TP_SHM", shm.c_str(), 1);
            setenv("TP_DEVICE", same ? "0" : std::to_string(r).c_str(), 1);
            setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);  // dmabuf IPC: what RCCL across processes needs on this driver stack
            execvp(argv[a], argv + a);
            perror("execvp");
            _exit(127);
        }
        kids.push_back(p);
    }
    int rc = 0;
    for (size_t done = 0; done < kids.size(); done++) {
        int st = 0;
        const pid_t p = wait(&st);
        const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
        if (code && !rc) {
            rc = code;
            for (pid_t k : kids)
                if (k != p) kill(k, SIGTERM);  // the others would only wait for the dead rank
        }
    }
    shm_unlink(shm.c_str());  // (rank 0 removes the name once everybody is attached; this is for jobs that never got there)
    return rc;
}

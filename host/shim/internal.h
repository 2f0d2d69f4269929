// shim/internal.h -- what the translation units of the PETSc-named shim share: the object layouts behind the opaque PETSc
// handles, the options database, the mesh / rank state of the process, vector mirrors, and the resolution of a KSP object
// graph into the library's solver options.  Header-only (inline functions and variables: one instance per process).
// The public surface is split by PETSc class: sys.cc (Petsc*, MPI_*, options, viewers), dmda.cc, vec.cc, mat.cc, ksp.cc, ext.cc.
#pragma once
//
// The shim: the PETSc 3.11 subset of include/petsc_compat/petsc.h on top of the C ABI of libtopopt_amd.so.
// Pure host code (g++): no HIP, no PETSc.  See the header for what is different behind the names: vectors live in
// HBM, MatSetValuesLocal is a capture (no matrix is ever assembled), the KSP/PCMG object graph is recorded and the
// configuration that is solved is CG + PCMG(V, Galerkin) + Chebyshev/Jacobi, selected through the options database
// exactly as with real PETSc (anything else: PETSC_ERR_SUP with a message, never a silent substitution).
#include <petsc/private/dmdaimpl.h>

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <chrono>
#include <cmath>
#include <map>
#include <string>
#include <vector>

#include "../../include/topopt_amd.h"
#include "../slab_comm.h"


namespace tpshim {

enum { CLS_DM = 1, CLS_VEC, CLS_MAT, CLS_KSP, CLS_PC, CLS_VIEWER, CLS_RANDOM, CLS_L2G };
struct Hdr {  // 32 bytes = the void *hdr_[4] of the public struct _p_DM
    int classid, refct;
    const char *type_name;
    void *r0, *r1;
};
static_assert(sizeof(Hdr) == 4 * sizeof(void *), "object header layout");
#pragma GCC visibility push(hidden)  // helpers and state below stay inside the library

inline int sup(const char *what) {
    fprintf(stderr, "[petsc-compat] PETSC_ERR_SUP: %s\n", what);
    return PETSC_ERR_SUP;
}

// ---- options database --------------------------------------------------------------------------------------
inline std::map<std::string, std::string> &opts() {
    static std::map<std::string, std::string> o;
    static bool env_done = false;
    if (!env_done) {
        env_done = true;
        if (const char *e = getenv("PETSC_OPTIONS")) {
            std::vector<std::string> tok;
            std::string cur;
            for (const char *p = e;; p++) {
                if (*p == ' ' || *p == '\t' || *p == 0) {
                    if (!cur.empty()) tok.push_back(cur);
                    cur.clear();
                    if (!*p) break;
                } else {
                    cur.push_back(*p);
                }
            }
            for (size_t i = 0; i < tok.size(); i++)
                if (tok[i][0] == '-' && tok[i].size() > 1 && !(tok[i][1] >= '0' && tok[i][1] <= '9')) {
                    const bool val = i + 1 < tok.size() && !(tok[i + 1][0] == '-' && tok[i + 1].size() > 1 &&
                                                              !(tok[i + 1][1] >= '0' && tok[i + 1][1] <= '9') && tok[i + 1][1] != '.');
                    o[tok[i].substr(1)] = val ? tok[i + 1] : "";
                    if (val) i++;
                }
        }
    }
    return o;
}
inline const std::string *opt_find(const char *pre, const char *name) {
    std::string key = (pre ? pre : "");
    key += (name[0] == '-' ? name + 1 : name);
    auto it = opts().find(key);
    return it == opts().end() ? nullptr : &it->second;
}

// ---- the mesh all DMs of a program live on (one process per GPU; z-slabs over the ranks of the job) ---------
struct Mesh {
    int nx = 0, ny = 0, nz = 0;  // nodes
    double box[6] = {0, 1, 0, 1, 0, 1};
    bool have_box = false;
    tp_grid *g = nullptr;
    int users = 0;
};
inline Mesh mesh;

// ---- the ranks of the job (host/slabrun sets TP_RANK / TP_NRANKS / TP_SHM / TP_DEVICE; absent: one rank).  The shared
// segment is attached at the first collective: by then the options database knows -nx / -ny, which size the mailboxes.
inline SlabComm sc;
inline bool sc_ready = false;
inline int job_rank() {
    static const int r = getenv("TP_RANK") ? atoi(getenv("TP_RANK")) : 0;
    return r;
}
inline int job_size() {
    static const int n = getenv("TP_NRANKS") ? atoi(getenv("TP_NRANKS")) : 1;
    return n < 1 ? 1 : n;
}
inline const std::string *opt_find(const char *pre, const char *name);
inline int comm_ready() {
    if (sc_ready) return 0;
    long nx = mesh.nx, ny = mesh.ny;
    if (nx == 0) {
        const std::string *ox = opt_find(nullptr, "nx"), *oy = opt_find(nullptr, "ny");
        nx = ox ? atol(ox->c_str()) : 65;  // TopOpt.cc:106-108 defaults
        ny = oy ? atol(oy->c_str()) : 33;
    }
    long cap = std::max(12L * nx * ny, 1L << 16);
    if (getenv("TP_SLAB_CAP")) cap = std::max(cap, atol(getenv("TP_SLAB_CAP")));
    if (slab_comm_join(&sc, cap)) return PETSC_ERR_LIB;  // (device buffers: ensure_grid -- MPI_* alone needs no GPU)
    sc_ready = true;
    return 0;
}


#pragma GCC visibility pop
}  // namespace tpshim
using namespace tpshim;

struct _p_Vec {
    Hdr h;
    // Layout.  A nodal vector is stored like the library stores it: this rank's z-slab WITH its ghost planes (n_alloc
    // entries, = PETSc's ghosted local numbering for stencil width 1); the GLOBAL vector's local part is the window
    // [off, off + n) of the owned planes, the LOCAL (ghosted) vector of DMCreateLocalVector is the whole array.  Element
    // vectors hold the owned elements only (off = 0, n = n_alloc).  One rank: off = 0, n = n_alloc = nglob.
    long n;        // length as PETSc sees it on this rank (VecGetLocalSize)
    long n_alloc;  // entries stored
    long off;      // first entry of the window
    long nglob;    // VecGetSize of a global vector
    long goff;     // global index of entry `off` (natural = PETSc ordering: slabs are contiguous in z)
    bool is_local; // sequential vector (ghosted local vector, coordinates): VecGetSize = n
    double *d;  // [dev]; NULL for a host-only vector (coordinates)
    std::vector<double> host;
    // Lazy coherence of the host mirror and the HBM array.  VecGetArray hands out host.data() and from then on the HOST
    // copy is the authoritative one (the reference also leaves arrays checked out for good: MMA.cc:549-550 gets p0/q0
    // and never restores them) until a device operation needs the vector (din/dinout push it) or overwrites it (dout).
    bool host_valid, dev_valid;
    DM dm;            // borrowed
};
struct DMFull : _p_DM {
    PetscInt M, N, P, dof, sw;
    int zkind;                 // partition in z: 0 like nodes (P - 1 = R e: rank 0 owns e + 1 planes, the others e), 1 like elements (P = R e)
    std::vector<PetscInt> lzv; // planes per rank
    double box[6];
    bool have_box;
    DM_DA da;
    Vec coords;
    PetscInt own[3];
    bool uses_grid;
};
static DMFull *F(DM d) { return static_cast<DMFull *>(d); }

enum MatKind { K_ELAST, K_HELM, K_CONE, K_TMAT, K_INTERP, K_EXT_ELAST, K_EXT_FILTER };
struct _p_Mat {
    Hdr h;
    MatKind kind;
    DM dm;  // borrowed (descriptor copied)
    long n_rows, n_cols;
    // capture state
    std::vector<double> ref;     // first block seen (576 / 64 / 8 values)
    std::vector<double> ref0;    // K_ELAST: rank 0's first block = the element matrix of the operator
    std::vector<double> E;       // K_ELAST: per element multiplier of `ref`
    long ncalls;
    long nverified = 0;          // TP_SHIM_VERIFY=1: element blocks checked entry by entry
    double coneR;                // K_CONE
    std::vector<int> hrow, hcol; // K_CONE, TP_SHIM_VERIFY=1: every entry the caller inserted
    std::vector<double> hval;
    bool assembled_since_setup;  // new values since the operator was last built
    Vec Nvec;                    // K_ELAST: copy of the Dirichlet vector
    bool have_bc;
    tp_elasticity *e;
    tp_filter *f;
    double *dE;  // [dev] element multipliers
    KSP ksp;     // borrowed back reference (KSPSetOperators)
    bool ext_assembled;
};
struct _p_PC {
    Hdr h;
    std::string type;
    int nlevels;
    std::vector<KSP> lev;  // [0] = coarse solve
    std::vector<Mat> interp;
    int mgtype, cycle, galerkin;
    KSP owner;
};
struct _p_KSP {
    Hdr h;
    std::string type, prefix;
    double rtol, atol, dtol;
    int maxits, restart;
    bool nonzero_guess, from_options;
    Mat A;
    PC pc;
    int its;
    double rnorm;
    bool is_sub;
};
struct _p_PetscViewer {
    Hdr h;
    FILE *fp;
    PetscFileMode mode;
    bool ascii;
    long long pos;  // binary: byte position of the next object (every rank keeps it; the ranks write their own parts)
};
struct _mpi_compat_file {
    FILE *fp;
    long long disp;           // byte displacement of the view
    int vec_block, vec_stride, vec_esize;  // vector filetype (0 = contiguous)
    long long pos;            // elements written since the view was set
};
struct _p_PetscRandom {
    Hdr h;
    uint64_t state;
};
struct _p_ISLocalToGlobalMapping {
    Hdr h;
};


namespace tpshim {
#pragma GCC visibility push(hidden)

inline Mat g_last_helm = nullptr;  // the Helmholtz matrix a later MatCreateAIJ'ed T belongs to (PDEFilter.cc:143-170)

inline void hdr_init(Hdr &h, int cls, const char *type) {
    h.classid = cls;
    h.refct = 1;
    h.type_name = type;
    h.r0 = h.r1 = nullptr;
}

inline int ensure_grid() {
    if (mesh.g) return 0;
    if (mesh.nx < 2) return PETSC_ERR_ORDER;
    tp_grid_opts o;
    memset(&o, 0, sizeof(o));
    o.nx = mesh.nx;
    o.ny = mesh.ny;
    o.nz = mesh.nz;
    o.hx = (mesh.box[1] - mesh.box[0]) / (mesh.nx - 1);
    o.hy = (mesh.box[3] - mesh.box[2]) / (mesh.ny - 1);
    o.hz = (mesh.box[5] - mesh.box[4]) / (mesh.nz - 1);
    int rc = comm_ready();
    if (rc) return rc;
    if (sc.nranks > 1 && sc.hooks.cap < std::max(3L * mesh.nx * mesh.ny, 4L * (mesh.nx - 1) * (mesh.ny - 1))) {
        fprintf(stderr, "[petsc-compat] the mailboxes of the job were sized before the mesh was known (%ld doubles): pass -nx/-ny or set TP_SLAB_CAP\n", sc.hooks.cap);
        return PETSC_ERR_LIB;
    }
    if (slab_comm_alloc(&sc)) return PETSC_ERR_LIB;
    o.rank = sc.rank;
    o.nranks = sc.nranks;
    o.device = sc.device;
    o.comm = sc.nranks > 1 ? &sc.hooks : nullptr;
    rc = tp_grid_create(&mesh.g, &o);
    if (rc) return rc;
    sc.grid = mesh.g;
    slab_comm_try_rccl(&sc, mesh.g);  // one process per GPU: the library's own RCCL path; else the mailboxes stay
    return 0;
}
inline bool is_nodal(const DMFull *d) { return d->M == mesh.nx && d->N == mesh.ny && d->P == mesh.nz; }
inline bool is_elem(const DMFull *d) { return d->M == mesh.nx - 1 && d->N == mesh.ny - 1 && d->P == mesh.nz - 1; }

// this rank's part of a DMDA along z (x and y are never split): owned range [zs, zs + zm), ghosted range [gzs, gzs + gzm)
// for stencil width sw (DMDAGetCorners / DMDAGetGhostCorners of a DM_BOUNDARY_NONE DMDA on a 1 x 1 x R process grid)
struct ZBox {
    PetscInt zs, zm, gzs, gzm;
};
inline ZBox zbox(const DMFull *d, PetscInt sw) {
    const int R = job_size(), r = job_rank();
    ZBox b;
    if (R == 1) {
        b.zs = b.gzs = 0;
        b.zm = b.gzm = d->P;
        return b;
    }
    if (d->zkind == 0) {
        const PetscInt e = (d->P - 1) / R;
        b.zs = r == 0 ? 0 : r * e + 1;
        b.zm = e + (r == 0 ? 1 : 0);
    } else {
        const PetscInt e = d->P / R;
        b.zs = r * e;
        b.zm = e;
    }
    b.gzs = std::max<PetscInt>(b.zs - sw, 0);
    b.gzm = std::min<PetscInt>(b.zs + b.zm + sw, d->P) - b.gzs;
    return b;
}

inline int vec_create_layout(long n_alloc, long off, long n, long nglob, long goff, bool is_local, bool host_only, DM dm, Vec *out);
inline int vec_create(long n, bool host_only, DM dm, Vec *out) {  // a vector that is not split (one rank, or sequential)
    return vec_create_layout(n, 0, n, n, 0, true, host_only, dm, out);
}
inline int vec_create_layout(long n_alloc, long off, long n, long nglob, long goff, bool is_local, bool host_only, DM dm, Vec *out) {
    Vec v = new _p_Vec();
    hdr_init(v->h, CLS_VEC, is_local ? "seq" : "mpi");
    v->n = n;
    v->n_alloc = n_alloc;
    v->off = off;
    v->nglob = nglob;
    v->goff = goff;
    v->is_local = is_local;
    v->d = nullptr;
    v->host_valid = false;
    v->dev_valid = true;
    v->dm = dm;
    if (host_only) {
        v->host.assign((size_t)n_alloc, 0.0);
        v->host_valid = true;
    } else {
        int rc = ensure_grid();
        if (!rc) rc = tp_malloc((void **)&v->d, sizeof(double) * (size_t)(n_alloc > 0 ? n_alloc : 1));
        if (!rc) rc = tp_vec_set(mesh.g, v->d, 0.0, n_alloc);
        if (rc) {
            delete v;
            return rc;
        }
    }
    *out = v;
    return 0;
}
inline int vec_pull(Vec x) {  // make the host mirror current
    if (!x->d || x->host_valid) return 0;
    x->host.resize((size_t)x->n_alloc);
    tp_sync(mesh.g);
    int rc = tp_memcpy_d2h(x->host.data(), x->d, sizeof(double) * (size_t)x->n_alloc);
    x->host_valid = rc == 0;
    return rc;
}
inline int vec_push(Vec x) {  // make the HBM array current
    if (!x->d || x->dev_valid) return 0;
    int rc = tp_memcpy_h2d(x->d, x->host.data(), sizeof(double) * (size_t)x->n_alloc);
    x->dev_valid = rc == 0;
    return rc;
}
// device pointers for an operation that reads / overwrites / updates the vector: the WINDOW PETSc sees (owned entries) ...
inline double *din(Vec x) {
    vec_push(x);
    return x->d + x->off;
}
inline double *dout(Vec x) {
    if (x->n != x->n_alloc) vec_push(x);  // the entries outside the window keep their values
    x->dev_valid = true;
    x->host_valid = false;
    return x->d + x->off;
}
inline double *dinout(Vec x) {
    vec_push(x);
    x->host_valid = false;
    return x->d + x->off;
}
// ... and the whole slab array, for the library calls that take nodal vectors with their ghost planes
inline double *bin(Vec x) { return din(x) - x->off; }
inline double *bout(Vec x) { return dout(x) - x->off; }
inline double *binout(Vec x) { return dinout(x) - x->off; }

// a Mat object of the given kind (no storage: MatSetValuesLocal is a capture, the operators live in the library)
inline Mat mat_new(MatKind kind, DM dm, long nr, long nc, const char *type) {
    Mat A = new _p_Mat();
    hdr_init(A->h, CLS_MAT, type);
    A->kind = kind;
    A->dm = dm;
    A->n_rows = nr;
    A->n_cols = nc;
    A->ncalls = 0;
    A->coneR = 0.0;
    A->assembled_since_setup = false;
    A->Nvec = nullptr;
    A->have_bc = false;
    A->e = nullptr;
    A->f = nullptr;
    A->dE = nullptr;
    A->ksp = nullptr;
    A->ext_assembled = false;
    return A;
}

// ---- the solver configuration a KSP resolves to ------------------------------------------------------------
inline void ksp_apply_options(KSP k, const std::vector<std::string> &prefixes) {
    for (const std::string &p : prefixes) {
        if (const std::string *v = opt_find(p.c_str(), "ksp_type")) k->type = *v;
        if (const std::string *v = opt_find(p.c_str(), "ksp_rtol")) k->rtol = atof(v->c_str());
        if (const std::string *v = opt_find(p.c_str(), "ksp_atol")) k->atol = atof(v->c_str());
        if (const std::string *v = opt_find(p.c_str(), "ksp_divtol")) k->dtol = atof(v->c_str());
        if (const std::string *v = opt_find(p.c_str(), "ksp_max_it")) k->maxits = atoi(v->c_str());
        if (const std::string *v = opt_find(p.c_str(), "ksp_gmres_restart")) k->restart = atoi(v->c_str());
        if (const std::string *v = opt_find(p.c_str(), "pc_type")) k->pc->type = *v;
    }
}
inline const char *NEED =
    "the MI355X path solves CG + PCMG(V-cycle, Galerkin) with Chebyshev/Jacobi smoothers (fast; any number of slabs) or "
    "FGMRES + PCMG with GMRES smoothers / coarse solve and SOR or Jacobi (the reference's hard-coded configuration, run as "
    "written on ONE device: a correctness mode); select the fast one like with real PETSc: -ksp_type cg "
    "-mg_levels_ksp_type chebyshev -mg_levels_pc_type jacobi -mg_coarse_ksp_type chebyshev -mg_coarse_pc_type jacobi "
    "(argv of PetscInitialize, $PETSC_OPTIONS or PetscOptionsSetValue)";

inline int resolve(KSP k, tp_solver_opts *o) {
    tp_solver_default_opts(o);
    // PETSc reads the level KSPs' options in PCSetUp_MG, AFTER the reference's hard-coded KSPSetType calls
    PC pc = k->pc;
    const int nl = pc->nlevels > 0 ? pc->nlevels : 1;
    for (int l = 0; l < (int)pc->lev.size(); l++) {
        std::vector<std::string> pre;
        if (l == 0 && nl > 1) {
            pre.push_back("mg_coarse_");
        } else {
            pre.push_back("mg_levels_");
            pre.push_back("mg_levels_" + std::to_string(l) + "_");
        }
        ksp_apply_options(pc->lev[l], pre);
    }
    const bool flexible = k->type == KSPFGMRES;  // LinearElasticity.cc:638, PDEFilter.cc:276
    if (k->type != KSPCG && !flexible) return sup((std::string("outer KSP type '") + k->type + "': " + NEED).c_str());
    if (pc->type != PCMG) return sup((std::string("PC type '") + pc->type + "': " + NEED).c_str());
    if (flexible && job_size() > 1)
        return sup((std::string("FGMRES + GMRES/SOR on more than one rank (its SOR is rank-local in PETSc: results depend on the "
                                "partition): ") + NEED).c_str());
    o->nlvls = nl;
    o->rtol = k->rtol;
    o->atol = k->atol;
    o->dtol = k->dtol;
    o->max_it = k->maxits;
    if (pc->mgtype != PC_MG_MULTIPLICATIVE) return sup("PCMG: only PC_MG_MULTIPLICATIVE");
    if (pc->cycle != PC_MG_CYCLE_V && pc->cycle != PC_MG_CYCLE_W) return sup("PCMG: cycle type neither V nor W");
    if (pc->cycle == PC_MG_CYCLE_W && flexible) return sup("PCMG: W-cycles with the FGMRES / GMRES level solvers");
    if (nl > 1 && pc->galerkin != PC_MG_GALERKIN_BOTH) return sup("PCMG: only -pc_mg_galerkin both");
    if (flexible) {
        // the configuration SetUpSolver hard-codes, as written (csrc/refksp.h)
        o->ksp_mode = 1;
        o->restart = k->restart;
        for (int l = 0; l < (int)pc->lev.size(); l++) {
            KSP s = pc->lev[l];
            const bool sor = s->pc->type == PCSOR;
            if (s->type != KSPGMRES || (!sor && s->pc->type != PCJACOBI))
                return sup((std::string("level ") + std::to_string(l) + " solver '" + s->type + "/" + s->pc->type +
                            "' under FGMRES (GMRES with SOR or Jacobi is what the reference sets): " + NEED).c_str());
            if ((l == 0 && nl > 1) || nl == 1) {
                o->ncoarse = s->maxits;
                o->coarse_restart = s->restart;
                o->coarse_rtol = s->rtol;
                o->coarse_pc = sor ? 1 : 0;
            } else {
                if (s->restart < s->maxits) return sup("level smoother: GMRES restart shorter than its iteration count");
                if (l > 1 && (o->nsmooth != s->maxits || o->smooth_pc != (sor ? 1 : 0)))
                    return sup("level smoothers that differ from level to level");
                o->nsmooth = s->maxits;
                o->smooth_pc = sor ? 1 : 0;
            }
        }
        return 0;
    }
    for (int l = 0; l < (int)pc->lev.size(); l++) {
        KSP s = pc->lev[l];
        if (s->type != KSPCHEBYSHEV || s->pc->type != PCJACOBI)
            return sup((std::string("level ") + std::to_string(l) + " smoother '" + s->type + "/" + s->pc->type + "': " + NEED).c_str());
        if (l == 0 && nl > 1) o->ncoarse = s->maxits;
        else o->nsmooth = s->maxits;
    }
    if (nl == 1 && !pc->lev.empty()) o->ncoarse = pc->lev[0]->maxits;
    return 0;
}

inline int ensure_elasticity(Mat A) {
    if (A->kind != K_ELAST) return PETSC_ERR_ARG_WRONG;
    if (A->ref.empty()) return PETSC_ERR_ORDER;
    DMFull *d = F(A->dm);
    const long nel = (long)(d->M - 1) * (d->N - 1) * ((d->P - 1) / job_size());  // this rank's elements
    if (!A->e) {
        tp_solver_opts o;
        if (A->ksp) {
            int rc = resolve(A->ksp, &o);
            if (rc) return rc;
        } else {
            tp_solver_default_opts(&o);
            o.nlvls = 1;
        }
        int rc = ensure_grid();
        // the element matrix every rank hands to the library is rank 0's first block; a rank's own multipliers (relative
        // to ITS first block) are rescaled by the ratio of the two
        A->ref0 = A->ref;
        if (!rc && job_size() > 1) {
            if (job_rank() != 0) std::fill(A->ref0.begin(), A->ref0.end(), 0.0);
            rc = MPI_Allreduce(A->ref0.data(), A->ref0.data(), 576, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
            const double f = A->ref[0] / A->ref0[0];
            for (int q : {1, 25, 300, 575})
                if (fabs(A->ref[q] - f * A->ref0[q]) > 1e-12 * fabs(f) * (fabs(A->ref0[0]) + fabs(A->ref0[q])))
                    return sup("dof-3 matrix: the ranks' element blocks are not multiples of one element matrix");
        }
        if (!rc) rc = tp_elasticity_create_ke(&A->e, mesh.g, &o, A->ref0.data());
        if (!rc && A->ksp && A->ksp->pc->cycle == PC_MG_CYCLE_W) {  // PCMGSetCycleType(pc, PC_MG_CYCLE_W): every level
            int two[TP_MAX_LEVELS];
            for (int &v : two) v = 2;
            rc = tp_elasticity_set_cycles(A->e, two, o.nlvls > 1 ? o.nlvls - 1 : 0);
        }
        if (rc) return rc;
        rc = tp_malloc((void **)&A->dE, sizeof(double) * (size_t)nel);
        if (rc) return rc;
        A->assembled_since_setup = true;
    }
    if (A->assembled_since_setup) {
        if (!A->have_bc || !A->Nvec) return sup("stiffness matrix without MatDiagonalScale(K, N, N): Dirichlet vector unknown");
        if ((long)A->E.size() != nel || A->ncalls != nel) return sup("MatSetValuesLocal: not every element was added exactly once");
        if (A->nverified) {
            printf("[petsc-compat] verified %ld element blocks (576 entries, 24 indices each) of the assembled stiffness matrix\n", A->nverified);
            A->nverified = 0;
        }
        // N arrived through window copies: its ghost planes are refreshed before the library reads the whole slab
        int rc = job_size() > 1 ? tp_grid_halo_nodes(mesh.g, binout(A->Nvec), 3) : 0;
        if (!rc) rc = tp_elasticity_set_bc(A->e, bin(A->Nvec));
        if (!rc) {
            const double f = A->ref[0] / A->ref0[0];
            if (f != 1.0) {
                std::vector<double> Es(A->E);
                for (double &v : Es) v *= f;
                rc = tp_memcpy_h2d(A->dE, Es.data(), sizeof(double) * (size_t)nel);
            } else {
                rc = tp_memcpy_h2d(A->dE, A->E.data(), sizeof(double) * (size_t)nel);
            }
        }
        // E_e = 0 + x^1 (1 - 0): the captured multipliers ARE the moduli (pow(x, 1.0) is exact)
        if (!rc) rc = tp_elasticity_assemble(A->e, A->dE, 0.0, 1.0, 1.0);
        if (rc) return rc;
        A->assembled_since_setup = false;
    }
    return 0;
}

inline int ensure_pdefilter(Mat K) {
    if (K->kind != K_HELM) return PETSC_ERR_ARG_WRONG;
    if (K->f) return 0;
    if (K->ref.size() != 64) return PETSC_ERR_ORDER;
    int rc = ensure_grid();
    if (rc) return rc;
    const double hx = (mesh.box[1] - mesh.box[0]) / (mesh.nx - 1), hy = (mesh.box[3] - mesh.box[2]) / (mesh.ny - 1),
                 hz = (mesh.box[5] - mesh.box[4]) / (mesh.nz - 1);
    // KF = R^2 int grad N . grad N + int N N  (PDEFilter.cc:476-565): recover R from the trace
    double tr = 0.0;
    for (int a = 0; a < 8; a++) tr += K->ref[9 * a];
    const double vol = hx * hy * hz;
    const double R2 = (tr - 8.0 * vol / 27.0) / (8.0 / 9.0 * (hy * hz / hx + hx * hz / hy + hx * hy / hz));
    if (!(R2 > 0)) return sup("8x8 blocks of the dof-1 matrix are not a Helmholtz filter element matrix");
    const double rmin = std::sqrt(R2) * 2.0 * std::sqrt(3.0);
    tp_solver_opts o;
    if (K->ksp) {
        rc = resolve(K->ksp, &o);
        if (rc) return rc;
    } else {
        return PETSC_ERR_ORDER;
    }
    rc = tp_filter_create(&K->f, mesh.g, 2, rmin, &o);
    if (rc) return rc;
    double kf[64];
    tp_filter_get_kf(K->f, kf);
    double scale = 0.0, dev = 0.0;
    for (int i = 0; i < 64; i++) {
        scale = fmax(scale, fabs(kf[i]));
        dev = fmax(dev, fabs(kf[i] - K->ref[i]));
    }
    if (dev > 1e-9 * scale) return sup("8x8 blocks of the dof-1 matrix differ from the Helmholtz element matrix of their own radius");
    return 0;
}

#pragma GCC visibility pop
}  // namespace tpshim

// petsc_shim.cc -- the PETSc 3.11 subset of include/petsc_compat/petsc.h on top of the C ABI of libtopopt_amd.so.
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

#include "../include/topopt_amd.h"
#include "slab_comm.h"

namespace {

enum { CLS_DM = 1, CLS_VEC, CLS_MAT, CLS_KSP, CLS_PC, CLS_VIEWER, CLS_RANDOM, CLS_L2G };
struct Hdr {  // 32 bytes = the void *hdr_[4] of the public struct _p_DM
    int classid, refct;
    const char *type_name;
    void *r0, *r1;
};
static_assert(sizeof(Hdr) == 4 * sizeof(void *), "object header layout");

int sup(const char *what) {
    fprintf(stderr, "[petsc-compat] PETSC_ERR_SUP: %s\n", what);
    return PETSC_ERR_SUP;
}

// ---- options database --------------------------------------------------------------------------------------
std::map<std::string, std::string> &opts() {
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
const std::string *opt_find(const char *pre, const char *name) {
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
} mesh;

// ---- the ranks of the job (host/slabrun sets TP_RANK / TP_NRANKS / TP_SHM / TP_DEVICE; absent: one rank).  The shared
// segment is attached at the first collective: by then the options database knows -nx / -ny, which size the mailboxes.
SlabComm sc;
bool sc_ready = false;
int job_rank() {
    static const int r = getenv("TP_RANK") ? atoi(getenv("TP_RANK")) : 0;
    return r;
}
int job_size() {
    static const int n = getenv("TP_NRANKS") ? atoi(getenv("TP_NRANKS")) : 1;
    return n < 1 ? 1 : n;
}
const std::string *opt_find(const char *pre, const char *name);
int comm_ready() {
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

}  // namespace

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

namespace {

Mat g_last_helm = nullptr;  // the Helmholtz matrix a later MatCreateAIJ'ed T belongs to (PDEFilter.cc:143-170)

void hdr_init(Hdr &h, int cls, const char *type) {
    h.classid = cls;
    h.refct = 1;
    h.type_name = type;
    h.r0 = h.r1 = nullptr;
}

int ensure_grid() {
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
bool is_nodal(const DMFull *d) { return d->M == mesh.nx && d->N == mesh.ny && d->P == mesh.nz; }
bool is_elem(const DMFull *d) { return d->M == mesh.nx - 1 && d->N == mesh.ny - 1 && d->P == mesh.nz - 1; }

// this rank's part of a DMDA along z (x and y are never split): owned range [zs, zs + zm), ghosted range [gzs, gzs + gzm)
// for stencil width sw (DMDAGetCorners / DMDAGetGhostCorners of a DM_BOUNDARY_NONE DMDA on a 1 x 1 x R process grid)
struct ZBox {
    PetscInt zs, zm, gzs, gzm;
};
ZBox zbox(const DMFull *d, PetscInt sw) {
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

int vec_create_layout(long n_alloc, long off, long n, long nglob, long goff, bool is_local, bool host_only, DM dm, Vec *out);
int vec_create(long n, bool host_only, DM dm, Vec *out) {  // a vector that is not split (one rank, or sequential)
    return vec_create_layout(n, 0, n, n, 0, true, host_only, dm, out);
}
int vec_create_layout(long n_alloc, long off, long n, long nglob, long goff, bool is_local, bool host_only, DM dm, Vec *out) {
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
int vec_pull(Vec x) {  // make the host mirror current
    if (!x->d || x->host_valid) return 0;
    x->host.resize((size_t)x->n_alloc);
    tp_sync(mesh.g);
    int rc = tp_memcpy_d2h(x->host.data(), x->d, sizeof(double) * (size_t)x->n_alloc);
    x->host_valid = rc == 0;
    return rc;
}
int vec_push(Vec x) {  // make the HBM array current
    if (!x->d || x->dev_valid) return 0;
    int rc = tp_memcpy_h2d(x->d, x->host.data(), sizeof(double) * (size_t)x->n_alloc);
    x->dev_valid = rc == 0;
    return rc;
}
// device pointers for an operation that reads / overwrites / updates the vector: the WINDOW PETSc sees (owned entries) ...
double *din(Vec x) {
    vec_push(x);
    return x->d + x->off;
}
double *dout(Vec x) {
    if (x->n != x->n_alloc) vec_push(x);  // the entries outside the window keep their values
    x->dev_valid = true;
    x->host_valid = false;
    return x->d + x->off;
}
double *dinout(Vec x) {
    vec_push(x);
    x->host_valid = false;
    return x->d + x->off;
}
// ... and the whole slab array, for the library calls that take nodal vectors with their ghost planes
double *bin(Vec x) { return din(x) - x->off; }
double *bout(Vec x) { return dout(x) - x->off; }
double *binout(Vec x) { return dinout(x) - x->off; }

// ---- the solver configuration a KSP resolves to ------------------------------------------------------------
void ksp_apply_options(KSP k, const std::vector<std::string> &prefixes) {
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
const char *NEED =
    "the MI355X path solves CG + PCMG(V-cycle, Galerkin) with Chebyshev/Jacobi smoothers (fast; any number of slabs) or "
    "FGMRES + PCMG with GMRES smoothers / coarse solve and SOR or Jacobi (the reference's hard-coded configuration, run as "
    "written on ONE device: a correctness mode); select the fast one like with real PETSc: -ksp_type cg "
    "-mg_levels_ksp_type chebyshev -mg_levels_pc_type jacobi -mg_coarse_ksp_type chebyshev -mg_coarse_pc_type jacobi "
    "(argv of PetscInitialize, $PETSC_OPTIONS or PetscOptionsSetValue)";

int resolve(KSP k, tp_solver_opts *o) {
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

int ensure_elasticity(Mat A) {
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

int ensure_pdefilter(Mat K) {
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

}  // namespace

extern "C" {

// =============================================================================================== Sys
PetscErrorCode PetscInitialize(int *argc, char ***args, const char[], const char[]) {
    if (argc && args)
        for (int i = 1; i < *argc; i++) {
            const char *a = (*args)[i];
            if (a[0] == '-' && a[1] && !(a[1] >= '0' && a[1] <= '9')) {
                const bool val = i + 1 < *argc && !((*args)[i + 1][0] == '-' && (*args)[i + 1][1] && !((*args)[i + 1][1] >= '0' && (*args)[i + 1][1] <= '9') && (*args)[i + 1][1] != '.');
                opts()[a + 1] = val ? (*args)[i + 1] : "";
                if (val) i++;
            }
        }
    return 0;
}
PetscErrorCode PetscFinalize(void) { return 0; }
PetscErrorCode PetscOptionsSetValue(PetscOptions, const char name[], const char value[]) {
    opts()[name[0] == '-' ? name + 1 : name] = value ? value : "";
    return 0;
}
PetscErrorCode PetscOptionsClearValue(PetscOptions, const char name[]) {
    opts().erase(name[0] == '-' ? name + 1 : name);
    return 0;
}
PetscErrorCode PetscOptionsGetInt(PetscOptions, const char pre[], const char name[], PetscInt *v, PetscBool *set) {
    const std::string *s = opt_find(pre, name);
    if (set) *set = s ? PETSC_TRUE : PETSC_FALSE;
    if (s) *v = atoi(s->c_str());
    return 0;
}
PetscErrorCode PetscOptionsGetReal(PetscOptions, const char pre[], const char name[], PetscReal *v, PetscBool *set) {
    const std::string *s = opt_find(pre, name);
    if (set) *set = s ? PETSC_TRUE : PETSC_FALSE;
    if (s) *v = atof(s->c_str());
    return 0;
}
PetscErrorCode PetscOptionsGetBool(PetscOptions, const char pre[], const char name[], PetscBool *v, PetscBool *set) {
    const std::string *s = opt_find(pre, name);
    if (set) *set = s ? PETSC_TRUE : PETSC_FALSE;
    if (s) *v = (s->empty() || *s == "1" || *s == "true" || *s == "yes" || *s == "TRUE") ? PETSC_TRUE : PETSC_FALSE;
    return 0;
}
PetscErrorCode PetscOptionsGetString(PetscOptions, const char pre[], const char name[], char str[], size_t len, PetscBool *set) {
    const std::string *s = opt_find(pre, name);
    if (set) *set = s ? PETSC_TRUE : PETSC_FALSE;
    if (s && len) {
        strncpy(str, s->c_str(), len - 1);
        str[len - 1] = 0;
    }
    return 0;
}
PetscErrorCode PetscPrintf(MPI_Comm comm, const char format[], ...) {
    if (comm == MPI_COMM_WORLD && job_rank() != 0) return 0;  // the first rank of the communicator prints
    va_list ap;
    va_start(ap, format);
    vprintf(format, ap);
    va_end(ap);
    fflush(stdout);
    return 0;
}
PetscErrorCode PetscErrorPrintf(const char format[], ...) {
    va_list ap;
    va_start(ap, format);
    vfprintf(stderr, format, ap);
    va_end(ap);
    return 0;
}
PetscErrorCode PetscMallocCompat(size_t n, void **p) {
    *p = malloc(n ? n : 1);
    return *p ? 0 : 55;
}
PetscErrorCode PetscFreeCompat(void *p) {
    free(p);
    return 0;
}
PetscErrorCode PetscObjectTypeCompare(PetscObject obj, const char type_name[], PetscBool *same) {
    const Hdr *h = (const Hdr *)obj;
    const char *t = h ? h->type_name : nullptr;
    if (h && h->classid == CLS_PC) t = ((PC)obj)->type.c_str();
    if (h && h->classid == CLS_KSP) t = ((KSP)obj)->type.c_str();
    *same = (t && type_name && strcmp(t, type_name) == 0) ? PETSC_TRUE : PETSC_FALSE;
    return 0;
}
static int mpi_esize(MPI_Datatype t) {
    switch (t) {
    case MPI_CHAR: return 1;
    case MPI_INT: case MPI_FLOAT: return 4;
    default: return 8;
    }
}
struct VecType {
    int count, block, stride, esize;
};
static std::vector<VecType> &vec_types() {
    static std::vector<VecType> v;
    return v;
}
// the element types of the reference's reductions as doubles and back (counts stay far below 2^53)
static double mpi_load(const void *p, int i, MPI_Datatype t) {
    switch (t) {
    case MPI_INT: return (double)((const int *)p)[i];
    case MPI_FLOAT: return (double)((const float *)p)[i];
    case MPI_UNSIGNED_LONG: return (double)((const unsigned long *)p)[i];
    case MPI_LONG: return (double)((const long *)p)[i];
    case MPI_CHAR: return (double)((const char *)p)[i];
    default: return ((const double *)p)[i];
    }
}
static void mpi_store(void *p, int i, MPI_Datatype t, double v) {
    switch (t) {
    case MPI_INT: ((int *)p)[i] = (int)v; break;
    case MPI_FLOAT: ((float *)p)[i] = (float)v; break;
    case MPI_UNSIGNED_LONG: ((unsigned long *)p)[i] = (unsigned long)v; break;
    case MPI_LONG: ((long *)p)[i] = (long)v; break;
    case MPI_CHAR: ((char *)p)[i] = (char)v; break;
    default: ((double *)p)[i] = v; break;
    }
}
int MPI_Allreduce(const void *s, void *r, int count, MPI_Datatype t, MPI_Op op, MPI_Comm comm) {
    if (comm == MPI_COMM_SELF || job_size() == 1) {
        if (s != r) memcpy(r, s, (size_t)count * (size_t)mpi_esize(t));
        return 0;
    }
    if (comm_ready()) return 1;
    const int how = op == MPI_SUM ? 0 : (op == MPI_MAX ? 1 : 2);
    for (int i0 = 0; i0 < count; i0 += 1024) {  // sums in rank order on every rank: the same bits everywhere
        const int c = std::min(1024, count - i0);
        double v[1024];
        for (int i = 0; i < c; i++) v[i] = mpi_load(s, i0 + i, t);
        slab_detail::host_reduce(&sc, v, c, how);
        for (int i = 0; i < c; i++) mpi_store(r, i0 + i, t, v[i]);
    }
    return 0;
}
int MPI_Allgather(const void *s, int scount, MPI_Datatype st, void *r, int, MPI_Datatype, MPI_Comm comm) {
    if (comm == MPI_COMM_SELF || job_size() == 1) {
        if (s != r) memcpy(r, s, (size_t)scount * (size_t)mpi_esize(st));
        return 0;
    }
    if (comm_ready() || scount > sc.hooks.cap) return 1;
    for (int i = 0; i < scount; i++) sc.mailbox(sc.rank, 0)[i] = mpi_load(s, i, st);
    sc.barrier();
    for (int q = 0; q < sc.nranks; q++)
        for (int i = 0; i < scount; i++) mpi_store(r, q * scount + i, st, sc.mailbox(q, 0)[i]);
    sc.barrier();
    return 0;
}
int MPI_Init(int *, char ***) { return 0; }
int MPI_Finalize(void) { return 0; }
int MPI_Abort(MPI_Comm, int code) {
    fprintf(stderr, "[mpi-compat] MPI_Abort(%d)\n", code);
    exit(code ? code : 1);
}
int MPI_Type_size(MPI_Datatype t, int *size) {
    *size = mpi_esize(t);
    return 0;
}
int MPI_Type_vector(int count, int blocklength, int stride, MPI_Datatype oldtype, MPI_Datatype *newtype) {
    vec_types().push_back({count, blocklength, stride, mpi_esize(oldtype)});
    *newtype = 1000 + (int)vec_types().size() - 1;
    return 0;
}
int MPI_Type_commit(MPI_Datatype *) { return 0; }
int MPI_Type_free(MPI_Datatype *t) {
    *t = 0;
    return 0;
}
int MPI_File_open(MPI_Comm, const char *filename, int amode, MPI_Info, MPI_File *fh) {
    // MPI-IO never truncates: several open/close rounds -- and several ranks, each writing through its own view --
    // build one file
    const int fd = open(filename, O_RDWR | ((amode & MPI_MODE_CREATE) ? O_CREAT : 0), 0644);
    FILE *fp = fd >= 0 ? fdopen(fd, "r+b") : nullptr;
    if (!fp) return 1;
    *fh = new _mpi_compat_file{fp, 0, 0, 0, 0, 0};
    return 0;
}
int MPI_File_close(MPI_File *fh) {
    if (fh && *fh) {
        fclose((*fh)->fp);
        delete *fh;
        *fh = nullptr;
    }
    return 0;
}
int MPI_File_delete(const char *filename, MPI_Info) { return remove(filename) ? 1 : 0; }
int MPI_File_set_view(MPI_File fh, MPI_Offset disp, MPI_Datatype, MPI_Datatype filetype, const char *, MPI_Info) {
    fh->disp = disp;
    fh->pos = 0;
    fh->vec_block = fh->vec_stride = fh->vec_esize = 0;
    if (filetype >= 1000) {
        const VecType &v = vec_types()[(size_t)(filetype - 1000)];
        fh->vec_block = v.block;
        fh->vec_stride = v.stride;
        fh->vec_esize = v.esize;
    }
    return 0;
}
int MPI_File_write(MPI_File fh, const void *buf, int count, MPI_Datatype t, MPI_Status *) {
    const int es = mpi_esize(t);
    const char *p = (const char *)buf;
    if (!fh->vec_block) {
        fseek(fh->fp, (long)(fh->disp + fh->pos * es), SEEK_SET);
        fwrite(p, (size_t)es, (size_t)count, fh->fp);
        fh->pos += count;
        return 0;
    }
    for (int done = 0; done < count;) {  // blocks of vec_block elements every vec_stride elements
        const long long blk = fh->pos / fh->vec_block, within = fh->pos % fh->vec_block;
        const int n = (int)std::min<long long>(fh->vec_block - within, count - done);
        fseek(fh->fp, (long)(fh->disp + (blk * fh->vec_stride + within) * es), SEEK_SET);
        fwrite(p + (size_t)done * es, (size_t)es, (size_t)n, fh->fp);
        done += n;
        fh->pos += n;
    }
    return 0;
}
int MPI_File_write_all(MPI_File fh, const void *buf, int count, MPI_Datatype t, MPI_Status *st) { return MPI_File_write(fh, buf, count, t, st); }
int MPI_Comm_rank(MPI_Comm comm, int *rank) {
    *rank = comm == MPI_COMM_SELF ? 0 : job_rank();
    return 0;
}
int MPI_Comm_size(MPI_Comm comm, int *size) {
    *size = comm == MPI_COMM_SELF ? 1 : job_size();
    return 0;
}
int MPI_Barrier(MPI_Comm comm) {
    if (comm == MPI_COMM_SELF || job_size() == 1) return 0;
    if (comm_ready()) return 1;
    sc.barrier();
    return 0;
}
double MPI_Wtime(void) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
PetscErrorCode PetscViewerBinaryOpen(MPI_Comm comm, const char name[], PetscFileMode mode, PetscViewer *v) {
    FILE *fp = nullptr;
    long long pos0 = 0;
    if (job_size() == 1 || comm == MPI_COMM_SELF) {
        fp = fopen(name, mode == FILE_MODE_READ ? "rb" : (mode == FILE_MODE_APPEND ? "ab" : "wb"));
    } else {  // rank 0 creates / truncates, then every rank has the file open for positioned writes of its own part
        if (mode != FILE_MODE_READ && job_rank() == 0) {
            FILE *t = fopen(name, mode == FILE_MODE_APPEND ? "ab" : "wb");
            if (t) fclose(t);
        }
        MPI_Barrier(comm);
        fp = fopen(name, mode == FILE_MODE_READ ? "rb" : "r+b");
        if (fp && mode == FILE_MODE_APPEND) {
            fseek(fp, 0, SEEK_END);
            pos0 = ftell(fp);
        }
    }
    if (!fp) return PETSC_ERR_FILE_OPEN;
    PetscViewer w = new _p_PetscViewer();
    hdr_init(w->h, CLS_VIEWER, PETSCVIEWERBINARY);
    w->fp = fp;
    w->mode = mode;
    w->ascii = false;
    w->pos = pos0;
    *v = w;
    return 0;
}
PetscErrorCode PetscViewerCreate(MPI_Comm, PetscViewer *v) {
    PetscViewer w = new _p_PetscViewer();
    hdr_init(w->h, CLS_VIEWER, PETSCVIEWERASCII);
    w->fp = nullptr;
    w->mode = FILE_MODE_WRITE;
    w->ascii = true;
    w->pos = 0;
    *v = w;
    return 0;
}
PetscErrorCode PetscViewerSetType(PetscViewer v, PetscViewerType type) {
    v->ascii = strcmp(type, PETSCVIEWERASCII) == 0;
    return 0;
}
PetscErrorCode PetscViewerFileSetMode(PetscViewer v, PetscFileMode mode) {
    v->mode = mode;
    return 0;
}
PetscErrorCode PetscViewerFileSetName(PetscViewer v, const char name[]) {
    if (v->fp) fclose(v->fp);
    if (v->ascii && v->mode != FILE_MODE_READ && job_rank() != 0) {  // an ASCII viewer prints from the first rank only
        v->fp = nullptr;
        return 0;
    }
    v->fp = fopen(name, v->mode == FILE_MODE_READ ? "r" : (v->mode == FILE_MODE_APPEND ? "a" : "w"));
    return v->fp ? 0 : PETSC_ERR_FILE_OPEN;
}
PetscErrorCode PetscViewerASCIIPrintf(PetscViewer v, const char format[], ...) {
    if (!v->fp && v->ascii && job_rank() != 0) return 0;
    if (!v->fp) return PETSC_ERR_ORDER;
    va_list ap;
    va_start(ap, format);
    vfprintf(v->fp, format, ap);
    va_end(ap);
    return 0;
}
PetscErrorCode PetscViewerDestroy(PetscViewer *v) {
    if (v && *v) {
        if ((*v)->fp) fclose((*v)->fp);
        delete *v;
        *v = nullptr;
    }
    return 0;
}
PetscErrorCode PetscRandomCreate(MPI_Comm, PetscRandom *r) {
    *r = new _p_PetscRandom();
    hdr_init((*r)->h, CLS_RANDOM, PETSCRAND48);
    (*r)->state = 0x1234ABCD330EULL;  // rand48 default seed
    return 0;
}
PetscErrorCode PetscRandomSetType(PetscRandom, PetscRandomType) { return 0; }
PetscErrorCode PetscRandomDestroy(PetscRandom *r) {
    if (r && *r) {
        delete *r;
        *r = nullptr;
    }
    return 0;
}

// =============================================================================================== DMDA
PetscErrorCode DMDACreate3d(MPI_Comm, DMBoundaryType, DMBoundaryType, DMBoundaryType, DMDAStencilType, PetscInt M,
                            PetscInt N, PetscInt P, PetscInt m, PetscInt n, PetscInt p, PetscInt dof, PetscInt s,
                            const PetscInt[], const PetscInt[], const PetscInt lz[], DM *da) {
    if (!da || M < 1 || N < 1 || P < 1 || dof < 1) return PETSC_ERR_ARG_OUTOFRANGE;
    // the process grid is 1 x 1 x R (z-slabs): PETSC_DECIDE resolves to it, anything else is refused
    const int R = job_size();
    if ((m != PETSC_DECIDE && m != 1) || (n != PETSC_DECIDE && n != 1) || (p != PETSC_DECIDE && p != R))
        return sup("DMDACreate3d: the ranks of the job form a 1 x 1 x R process grid (z-slabs)");
    int zkind = 0;
    if (R > 1) {
        if ((P - 1) % R == 0) zkind = 0;
        else if (P % R == 0) zkind = 1;
        else return sup("DMDACreate3d: the z extent does not split into equal slabs of elements over the ranks");
        const PetscInt e = zkind == 0 ? (P - 1) / R : P / R;
        if (lz)
            for (int q = 0; q < R; q++)
                if (lz[q] != e + ((zkind == 0 && q == 0) ? 1 : 0)) return sup("DMDACreate3d: lz[] is not the slab partition of the node mesh");
    }
    DMFull *d = new DMFull();
    d->zkind = zkind;
    for (int q = 0; q < R; q++) d->lzv.push_back(R == 1 ? P : (zkind == 0 ? (P - 1) / R + (q == 0 ? 1 : 0) : P / R));
    Hdr h;
    hdr_init(h, CLS_DM, "da");
    memcpy(d->hdr_, &h, sizeof(h));
    d->data = &d->da;
    d->da.e = nullptr;
    d->da.ne = 0;
    d->da.elementtype = DMDA_ELEMENT_P1;  // PETSc's default; the reference sets Q1 itself
    d->M = M;
    d->N = N;
    d->P = P;
    d->dof = dof;
    d->sw = s;
    d->have_box = false;
    const double b[6] = {0, 1, 0, 1, 0, 1};
    memcpy(d->box, b, sizeof(b));
    d->coords = nullptr;
    d->own[0] = M;
    d->own[1] = N;
    d->own[2] = P;
    d->uses_grid = false;
    if (mesh.nx == 0) {  // the first DMDA of a program is the node mesh (TopOpt.cc:225-262)
        mesh.nx = M;
        mesh.ny = N;
        mesh.nz = P;
    }
    mesh.users++;
    *da = d;
    return 0;
}
PetscErrorCode DMSetFromOptions(DM) { return 0; }
PetscErrorCode DMSetUp(DM) { return 0; }
PetscErrorCode DMDASetUniformCoordinates(DM da, PetscReal x0, PetscReal x1, PetscReal y0, PetscReal y1, PetscReal z0,
                                         PetscReal z1) {
    DMFull *d = F(da);
    const double b[6] = {x0, x1, y0, y1, z0, z1};
    memcpy(d->box, b, sizeof(b));
    d->have_box = true;
    if (d->coords) {
        VecDestroy(&d->coords);
    }
    if (is_nodal(d) && !mesh.have_box && !mesh.g) {  // element size of the mesh: first nodal DM with coordinates
        memcpy(mesh.box, b, sizeof(b));
        mesh.have_box = true;
    }
    return 0;
}
PetscErrorCode DMDASetElementType(DM da, DMDAElementType t) {
    F(da)->da.elementtype = t;
    return 0;
}
PetscErrorCode DMDAGetInfo(DM da, PetscInt *dim, PetscInt *M, PetscInt *N, PetscInt *P, PetscInt *m, PetscInt *n,
                           PetscInt *p, PetscInt *dof, PetscInt *s, DMBoundaryType *bx, DMBoundaryType *by,
                           DMBoundaryType *bz, DMDAStencilType *st) {
    DMFull *d = F(da);
    if (dim) *dim = 3;
    if (M) *M = d->M;
    if (N) *N = d->N;
    if (P) *P = d->P;
    if (m) *m = 1;
    if (n) *n = 1;
    if (p) *p = job_size();
    if (dof) *dof = d->dof;
    if (s) *s = d->sw;
    if (bx) *bx = DM_BOUNDARY_NONE;
    if (by) *by = DM_BOUNDARY_NONE;
    if (bz) *bz = DM_BOUNDARY_NONE;
    if (st) *st = DMDA_STENCIL_BOX;
    return 0;
}
PetscErrorCode DMDAGetCorners(DM da, PetscInt *x, PetscInt *y, PetscInt *z, PetscInt *m, PetscInt *n, PetscInt *p) {
    DMFull *d = F(da);
    const ZBox b = zbox(d, d->sw);
    if (x) *x = 0;
    if (y) *y = 0;
    if (z) *z = b.zs;
    if (m) *m = d->M;
    if (n) *n = d->N;
    if (p) *p = b.zm;
    return 0;
}
PetscErrorCode DMDAGetGhostCorners(DM da, PetscInt *x, PetscInt *y, PetscInt *z, PetscInt *m, PetscInt *n, PetscInt *p) {
    DMFull *d = F(da);  // non-periodic, x and y unsplit: ghost points only towards the neighbouring slabs
    const ZBox b = zbox(d, d->sw);
    if (x) *x = 0;
    if (y) *y = 0;
    if (z) *z = b.gzs;
    if (m) *m = d->M;
    if (n) *n = d->N;
    if (p) *p = b.gzm;
    return 0;
}
PetscErrorCode DMDAGetOwnershipRanges(DM da, const PetscInt *lx[], const PetscInt *ly[], const PetscInt *lz[]) {
    DMFull *d = F(da);
    if (lx) *lx = &d->own[0];
    if (ly) *ly = &d->own[1];
    if (lz) *lz = d->lzv.data();
    return 0;
}
PetscErrorCode DMDAGetLocalInfo(DM da, DMDALocalInfo *i) {
    DMFull *d = F(da);
    memset(i, 0, sizeof(*i));
    i->dim = 3;
    i->dof = d->dof;
    i->sw = d->sw;
    const ZBox b = zbox(d, d->sw);
    i->mx = i->xm = i->gxm = d->M;
    i->my = i->ym = i->gym = d->N;
    i->mz = d->P;
    i->zs = b.zs;
    i->zm = b.zm;
    i->gzs = b.gzs;
    i->gzm = b.gzm;
    i->st = DMDA_STENCIL_BOX;
    i->da = da;
    return 0;
}
PetscErrorCode DMDAGetElements(DM da, PetscInt *nel, PetscInt *nen, const PetscInt *e[]) {
    DMFull *d = F(da);
    if (!d->da.e) {  // hexahedra, DMDA natural order (the numbering of LinearElasticity.cc:819-826), ghosted local node numbers
        const ZBox b = zbox(d, d->sw);
        const PetscInt ex = d->M - 1, ey = d->N - 1;
        const PetscInt k0 = (b.zs != b.gzs ? b.zs - 1 : b.zs) - b.gzs, ez = k0 + (b.zs + b.zm - 1 - (b.zs != b.gzs ? b.zs - 1 : b.zs));
        d->da.ne = ex * ey * (ez - k0);
        d->da.e = (PetscInt *)malloc(sizeof(PetscInt) * (size_t)(1 + 8 * (long)d->da.ne));
        long c = 0;
        for (PetscInt k = k0; k < ez; k++)
            for (PetscInt j = 0; j < ey; j++)
                for (PetscInt i = 0; i < ex; i++) {
                    const PetscInt n0 = i + d->M * (j + d->N * k), dz = d->M * d->N;
                    const PetscInt cell[8] = {n0, n0 + 1, n0 + 1 + d->M, n0 + d->M, n0 + dz, n0 + 1 + dz, n0 + 1 + d->M + dz, n0 + d->M + dz};
                    for (int q = 0; q < 8; q++) d->da.e[c++] = cell[q];
                }
    }
    *nel = d->da.ne;
    *nen = 8;
    *e = d->da.e;
    return 0;
}
PetscErrorCode DMDARestoreElements(DM, PetscInt *, PetscInt *, const PetscInt *[]) { return 0; }
PetscErrorCode DMGetCoordinatesLocal(DM da, Vec *c) {
    DMFull *d = F(da);
    if (!d->coords) {  // the ghosted local box of this rank
        const ZBox b = zbox(d, d->sw);
        const long n = (long)d->M * d->N * b.gzm;
        int rc = vec_create(3 * n, true, da, &d->coords);
        if (rc) return rc;
        const double hx = d->M > 1 ? (d->box[1] - d->box[0]) / (d->M - 1) : 0.0, hy = d->N > 1 ? (d->box[3] - d->box[2]) / (d->N - 1) : 0.0,
                     hz = d->P > 1 ? (d->box[5] - d->box[4]) / (d->P - 1) : 0.0;
        double *p = d->coords->host.data();
        for (PetscInt k = b.gzs; k < b.gzs + b.gzm; k++)
            for (PetscInt j = 0; j < d->N; j++)
                for (PetscInt i = 0; i < d->M; i++) {  // DMDASetUniformCoordinates: xmin + i * h
                    *p++ = d->box[0] + hx * i;
                    *p++ = d->box[2] + hy * j;
                    *p++ = d->box[4] + hz * k;
                }
    }
    *c = d->coords;
    return 0;
}
PetscErrorCode DMGetLocalToGlobalMapping(DM, ISLocalToGlobalMapping *m) {
    static _p_ISLocalToGlobalMapping identity;
    *m = &identity;
    return 0;
}
static int dm_vector(DM da, bool local, Vec *v) {
    DMFull *d = F(da);
    if (!is_nodal(d) && !is_elem(d)) return sup("vector on a DMDA that is neither the node mesh nor its element mesh");
    d->uses_grid = true;
    const long per = (long)d->dof * d->M * d->N;  // entries per z-plane
    const long nglob = per * d->P;
    if (job_size() == 1) return vec_create_layout(nglob, 0, nglob, nglob, 0, local, false, da, v);
    const ZBox own = zbox(d, 0);
    if (is_elem(d)) {
        if (local) return sup("DMCreateLocalVector on the element mesh across ranks");
        return vec_create_layout(per * own.zm, 0, per * own.zm, nglob, per * own.zs, false, false, da, v);
    }
    const ZBox gb = zbox(d, 1);  // the library's slab layout: one ghost plane towards each neighbour
    if (d->sw != 1) return sup("node mesh with a stencil width other than 1 across ranks");
    if (local) return vec_create_layout(per * gb.gzm, 0, per * gb.gzm, nglob, per * gb.gzs, true, false, da, v);
    return vec_create_layout(per * gb.gzm, per * (own.zs - gb.gzs), per * own.zm, nglob, per * own.zs, false, false, da, v);
}
PetscErrorCode DMCreateGlobalVector(DM da, Vec *v) { return dm_vector(da, false, v); }
PetscErrorCode DMCreateLocalVector(DM da, Vec *v) { return dm_vector(da, true, v); }
// global -> ghosted local: both are slab arrays; copy the slab, then fetch the ghost planes from the neighbours
PetscErrorCode DMGlobalToLocalBegin(DM da, Vec g, InsertMode, Vec l) {
    if (g == l) return 0;
    if (g->n_alloc != l->n_alloc) return PETSC_ERR_ARG_WRONG;
    const double *pg = bin(g);
    int rc = tp_vec_axpby(mesh.g, bout(l), 1.0, pg, 0.0, l->n_alloc);
    if (!rc && job_size() > 1) rc = tp_grid_halo_nodes(mesh.g, l->d, (int)F(da)->dof);
    return rc;
}
PetscErrorCode DMGlobalToLocalEnd(DM, Vec, InsertMode, Vec) { return 0; }
PetscErrorCode DMCoarsenHierarchy(DM da, PetscInt nlevels, DM dac[]) {
    DMFull *f = F(da);
    PetscInt M = f->M, N = f->N, P = f->P;
    for (PetscInt l = 0; l < nlevels; l++) {
        if ((M - 1) % 2 || (N - 1) % 2 || (P - 1) % 2) return sup("DMCoarsenHierarchy: element counts not divisible by 2 (TopOpt.cc:183-201)");
        M = (M - 1) / 2 + 1;
        N = (N - 1) / 2 + 1;
        P = (P - 1) / 2 + 1;
        int rc = DMDACreate3d(0, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DMDA_STENCIL_BOX, M, N, P, 1, 1, job_size(), f->dof, f->sw, 0, 0, 0, &dac[l]);
        if (rc) return rc;
    }
    return 0;
}
static Mat mat_new(MatKind kind, DM dm, long nr, long nc, const char *type) {
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
PetscErrorCode DMCreateInterpolation(DM dac, DM daf, Mat *P, Vec *scale) {
    DMFull *c = F(dac), *f = F(daf);
    if ((f->M - 1) != 2 * (c->M - 1) || (f->N - 1) != 2 * (c->N - 1) || (f->P - 1) != 2 * (c->P - 1))
        return sup("DMCreateInterpolation: only factor-2 trilinear (Q1) interpolation between DMDAs");
    *P = mat_new(K_INTERP, daf, (long)f->dof * f->M * f->N * zbox(f, 0).zm, (long)c->dof * c->M * c->N * zbox(c, 0).zm, "q1interp");
    if (scale) *scale = nullptr;
    return 0;
}
PetscErrorCode DMCreateMatrix(DM da, Mat *A) {
    DMFull *d = F(da);
    const long n = (long)d->dof * d->M * d->N * zbox(d, 0).zm;  // local rows
    if (is_nodal(d) && d->dof == 3) {
        *A = mat_new(K_ELAST, da, n, n, "topopt-elasticity");
    } else if (is_nodal(d) && d->dof == 1) {
        *A = mat_new(K_HELM, da, n, n, "topopt-helmholtz");
        g_last_helm = *A;
    } else if (is_elem(d) && d->dof == 1) {
        *A = mat_new(K_CONE, da, n, n, "topopt-conefilter");
    } else {
        return sup("DMCreateMatrix: dof-3 / dof-1 node mesh or dof-1 element mesh only");
    }
    return 0;
}
PetscErrorCode DMDestroy(DM *da) {
    if (da && *da) {
        DMFull *d = F(*da);
        if (d->coords) VecDestroy(&d->coords);
        free(d->da.e);
        delete d;
        *da = nullptr;
        if (--mesh.users == 0 && mesh.g) {
            tp_grid_destroy(mesh.g);
            mesh = Mesh();
        }
    }
    return 0;
}

// =============================================================================================== Vec
PetscErrorCode VecDuplicate(Vec v, Vec *nv) {
    return vec_create_layout(v->n_alloc, v->off, v->n, v->nglob, v->goff, v->is_local, v->d == nullptr, v->dm, nv);
}
PetscErrorCode VecDuplicateVecs(Vec v, PetscInt m, Vec *V[]) {
    *V = (Vec *)malloc(sizeof(Vec) * (size_t)(m > 0 ? m : 1));
    for (PetscInt i = 0; i < m; i++) {
        int rc = VecDuplicate(v, &(*V)[i]);
        if (rc) return rc;
    }
    return 0;
}
PetscErrorCode VecDestroyVecs(PetscInt m, Vec *V[]) {
    if (V && *V) {
        for (PetscInt i = 0; i < m; i++) VecDestroy(&(*V)[i]);
        free(*V);
        *V = nullptr;
    }
    return 0;
}
PetscErrorCode VecDestroy(Vec *v) {
    if (v && *v) {
        if (--(*v)->h.refct <= 0) {
            if ((*v)->d) tp_free((*v)->d);
            delete *v;
        }
        *v = nullptr;
    }
    return 0;
}
PetscErrorCode VecSet(Vec v, PetscScalar a) {
    if (!v->d) {
        std::fill(v->host.begin(), v->host.end(), a);
        return 0;
    }
    return tp_vec_set(mesh.g, dout(v), a, v->n);
}
PetscErrorCode VecCopy(Vec x, Vec y) {
    if (x->n != y->n) return PETSC_ERR_ARG_WRONG;
    if (x == y) return 0;
    const double *px = din(x);
    return tp_vec_axpby(mesh.g, dout(y), 1.0, px, 0.0, y->n);
}
PetscErrorCode VecScale(Vec v, PetscScalar a) { return tp_vec_scale(mesh.g, dinout(v), a, v->n); }
PetscErrorCode VecAXPY(Vec y, PetscScalar a, Vec x) {
    if (x->n != y->n) return PETSC_ERR_ARG_WRONG;
    const double *px = din(x);
    return tp_vec_axpby(mesh.g, dinout(y), a, px, 1.0, y->n);
}
PetscErrorCode VecAXPBY(Vec y, PetscScalar a, PetscScalar b, Vec x) {
    if (x->n != y->n) return PETSC_ERR_ARG_WRONG;
    const double *px = din(x);
    return tp_vec_axpby(mesh.g, dinout(y), a, px, b, y->n);
}
PetscErrorCode VecAXPBYPCZ(Vec z, PetscScalar alpha, PetscScalar beta, PetscScalar gamma, Vec x, Vec y) {  // z = a x + b y + c z
    const double *px = din(x), *py = din(y);
    double *pz = gamma == 0.0 ? dout(z) : dinout(z);
    int rc = tp_vec_axpby(mesh.g, pz, alpha, px, gamma, z->n);
    return rc ? rc : tp_vec_axpby(mesh.g, pz, beta, py, 1.0, z->n);
}
PetscErrorCode VecPointwiseMult(Vec w, Vec x, Vec y) {
    if (w->n != x->n || w->n != y->n) return PETSC_ERR_ARG_WRONG;
    const double *px = din(x), *py = din(y);
    return tp_vec_pointwise(mesh.g, (w == x || w == y) ? dinout(w) : dout(w), px, py, 0, w->n);
}
PetscErrorCode VecPointwiseDivide(Vec w, Vec x, Vec y) {
    if (w->n != x->n || w->n != y->n) return PETSC_ERR_ARG_WRONG;
    const double *px = din(x), *py = din(y);
    return tp_vec_pointwise(mesh.g, (w == x || w == y) ? dinout(w) : dout(w), px, py, 1, w->n);
}
PetscErrorCode VecDot(Vec x, Vec y, PetscScalar *val) {
    if (x->n != y->n) return PETSC_ERR_ARG_WRONG;
    return tp_vec_dot(mesh.g, din(x), din(y), x->n, val);
}
PetscErrorCode VecNorm(Vec x, NormType type, PetscReal *val) {
    if (type != NORM_2) return sup("VecNorm: NORM_2 only");
    double s = 0.0;
    const double *px = din(x);
    int rc = tp_vec_dot(mesh.g, px, px, x->n, &s);
    *val = std::sqrt(s);
    return rc;
}
PetscErrorCode VecSum(Vec x, PetscScalar *sum) { return tp_vec_dot(mesh.g, din(x), nullptr, x->n, sum); }
static int vec_extreme(Vec x, bool want_max, PetscInt *p, PetscReal *val) {
    int rc = vec_pull(x);
    const double *h = x->host.data() + x->off;
    long at = 0;
    for (long i = 1; i < x->n; i++)
        if (want_max ? h[i] > h[at] : h[i] < h[at]) at = i;
    double v = x->n ? h[at] : (want_max ? -1e300 : 1e300);
    if (!x->is_local && job_size() > 1) {
        if (p) return sup("VecMax / VecMin: the location of the extremum across ranks");
        if (comm_ready()) return PETSC_ERR_LIB;
        slab_detail::host_reduce(&sc, &v, 1, want_max ? 1 : 2);
    }
    if (p) *p = (PetscInt)at;
    if (val) *val = v;
    return rc;
}
PetscErrorCode VecMax(Vec x, PetscInt *p, PetscReal *val) { return vec_extreme(x, true, p, val); }
PetscErrorCode VecMin(Vec x, PetscInt *p, PetscReal *val) { return vec_extreme(x, false, p, val); }
PetscErrorCode VecGetSize(Vec x, PetscInt *n) {
    *n = (PetscInt)(x->is_local ? x->n : x->nglob);
    return 0;
}
PetscErrorCode VecGetLocalSize(Vec x, PetscInt *n) {
    *n = (PetscInt)x->n;
    return 0;
}
PetscErrorCode VecGetArray(Vec x, PetscScalar **a) {
    int rc = vec_pull(x);
    if (x->d) x->dev_valid = false;  // the caller may write through the pointer, now or later
    *a = x->host.data() + x->off;
    return rc;
}
PetscErrorCode VecRestoreArray(Vec, PetscScalar **a) {  // nothing to copy: the next device use pushes the mirror
    if (a) *a = nullptr;
    return 0;
}
PetscErrorCode VecGetArrays(const Vec x[], PetscInt n, PetscScalar **a[]) {
    PetscScalar **q = (PetscScalar **)malloc(sizeof(PetscScalar *) * (size_t)(n > 0 ? n : 1));
    for (PetscInt i = 0; i < n; i++) {
        int rc = VecGetArray(x[i], &q[i]);
        if (rc) return rc;
    }
    *a = q;
    return 0;
}
PetscErrorCode VecRestoreArrays(const Vec[], PetscInt, PetscScalar **a[]) {
    if (a && *a) {
        free(*a);
        *a = nullptr;
    }
    return 0;
}
// local numbering = PETSc's ghosted local numbering of the DMDA = the position in the stored slab.  An entry set on a
// ghost node stays local: the owner sets the same entry itself (the reference's set-up loops run over all local nodes
// on every rank, LinearElasticity.cc:148-172), which is what VecAssembly would deliver.
PetscErrorCode VecSetValueLocal(Vec v, PetscInt row, PetscScalar value, InsertMode mode) {
    if (row < 0 || row >= v->n_alloc) return PETSC_ERR_ARG_OUTOFRANGE;
    int rc = vec_pull(v);
    if (rc) return rc;
    if (v->d) v->dev_valid = false;
    if (mode == ADD_VALUES) v->host[(size_t)row] += value;
    else v->host[(size_t)row] = value;
    return 0;
}
PetscErrorCode VecSetValue(Vec v, PetscInt row, PetscScalar value, InsertMode mode) {  // global index
    if (row < v->goff || row >= v->goff + v->n) return job_size() > 1 ? sup("VecSetValue on an entry of another rank") : PETSC_ERR_ARG_OUTOFRANGE;
    return VecSetValueLocal(v, (PetscInt)(row - v->goff + v->off), value, mode);
}
PetscErrorCode VecAssemblyBegin(Vec) { return 0; }
PetscErrorCode VecAssemblyEnd(Vec v) { return vec_push(v); }
PetscErrorCode VecSetRandom(Vec v, PetscRandom r) {
    int rc0 = vec_pull(v);
    if (rc0) return rc0;
    // drand48's linear congruential generator; one global sequence in natural order whatever the partition
    for (long i = 0; i < v->nglob; i++) {
        r->state = (r->state * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
        if (i >= v->goff && i < v->goff + v->n) v->host[(size_t)(i - v->goff + v->off)] = (double)r->state / (double)(1ULL << 48);
    }
    v->host_valid = true;
    v->dev_valid = false;
    return vec_push(v);
}
// PETSc binary Vec: big-endian int32 class id 1211214, int32 n, n big-endian doubles
PetscErrorCode VecView(Vec v, PetscViewer w) {
    int rc = vec_pull(v);
    if (rc) return rc;
    auto be32 = [&](uint32_t x) {
        unsigned char b[4] = {(unsigned char)(x >> 24), (unsigned char)(x >> 16), (unsigned char)(x >> 8), (unsigned char)x};
        fwrite(b, 1, 4, w->fp);
    };
    // every rank writes its own part at its place (natural ordering = the slab order); rank 0 also the header
    const long ng = v->is_local ? v->n : v->nglob, g0 = v->is_local ? 0 : v->goff;
    fseek(w->fp, (long)w->pos, SEEK_SET);
    if (job_rank() == 0 || v->is_local) {
        be32(1211214u);
        be32((uint32_t)ng);
    }
    fseek(w->fp, (long)(w->pos + 8 + 8 * g0), SEEK_SET);
    for (long i = 0; i < v->n; i++) {
        uint64_t u;
        memcpy(&u, &v->host[(size_t)(v->off + i)], 8);
        unsigned char b[8];
        for (int k = 0; k < 8; k++) b[k] = (unsigned char)(u >> (56 - 8 * k));
        fwrite(b, 1, 8, w->fp);
    }
    fflush(w->fp);
    w->pos += 8 + 8 * ng;
    return 0;
}
PetscErrorCode VecLoad(Vec v, PetscViewer w) {
    unsigned char b[8];
    auto be32 = [&](uint32_t *x) {
        if (fread(b, 1, 4, w->fp) != 4) return false;
        *x = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
        return true;
    };
    uint32_t cls, n;
    const long ng = v->is_local ? v->n : v->nglob, g0 = v->is_local ? 0 : v->goff;
    fseek(w->fp, (long)w->pos, SEEK_SET);
    if (!be32(&cls) || !be32(&n) || cls != 1211214u || (long)n != ng) return 79;  // PETSC_ERR_FILE_UNEXPECTED
    int rc0 = vec_pull(v);
    if (rc0) return rc0;
    fseek(w->fp, (long)(w->pos + 8 + 8 * g0), SEEK_SET);
    for (long i = 0; i < v->n; i++) {
        if (fread(b, 1, 8, w->fp) != 8) return 79;
        uint64_t u = 0;
        for (int k = 0; k < 8; k++) u = (u << 8) | b[k];
        memcpy(&v->host[(size_t)(v->off + i)], &u, 8);
    }
    w->pos += 8 + 8 * ng;
    v->host_valid = true;
    v->dev_valid = false;
    return vec_push(v);
}
PetscErrorCode VecTopOptGetDevicePointer(Vec x, PetscScalar **d) {
    *d = dinout(x);
    return 0;
}

// =============================================================================================== Mat
PetscErrorCode MatCreateAIJ(MPI_Comm, PetscInt m, PetscInt n, PetscInt, PetscInt, PetscInt, const PetscInt[], PetscInt,
                            const PetscInt[], Mat *A) {
    // the only AIJ matrix of the path: T (nodes x elements, PDEFilter.cc:143-170)
    // local sizes: this rank's owned nodes x owned elements
    const int R = job_size(), r = job_rank();
    const long ezl = (mesh.nz - 1) / R, own_planes = R == 1 ? mesh.nz : ezl + (r == 0 ? 1 : 0);
    if (m != (PetscInt)((long)mesh.nx * mesh.ny * own_planes) || n != (PetscInt)((long)(mesh.nx - 1) * (mesh.ny - 1) * ezl))
        return sup("MatCreateAIJ: only the nodes x elements transfer matrix of the PDE filter");
    *A = mat_new(K_TMAT, nullptr, m, n, "topopt-elem2node");
    return 0;
}
PetscErrorCode MatSetLocalToGlobalMapping(Mat, ISLocalToGlobalMapping, ISLocalToGlobalMapping) { return 0; }
PetscErrorCode MatZeroEntries(Mat A) {
    A->ncalls = 0;
    if (A->kind == K_ELAST) std::fill(A->E.begin(), A->E.end(), 0.0);
    return 0;
}
PetscErrorCode MatSetValuesLocal(Mat A, PetscInt nrow, const PetscInt irow[], PetscInt ncol, const PetscInt icol[],
                                 const PetscScalar y[], InsertMode addv) {
    switch (A->kind) {
    case K_ELAST: {  // AssembleStiffnessMatrix, LinearElasticity.cc:510-524: ke = KE * dens, ADD_VALUES
        if (nrow != 24 || ncol != 24 || addv != ADD_VALUES) return sup("dof-3 matrix: 24x24 ADD_VALUES element blocks only");
        DMFull *d = F(A->dm);
        // local numbering: node planes counted from the first stored (ghost) plane, whose element layer is this rank's first
        const PetscInt ex = d->M - 1, ey = d->N - 1, ez = (d->P - 1) / job_size();
        const PetscInt n0 = irow[0] / 3, i = n0 % d->M, j = (n0 / d->M) % d->N, k = n0 / (d->M * d->N);
        if (irow[0] % 3 || i >= ex || j >= ey || k >= ez || icol[0] != irow[0] || irow[3] != 3 * (n0 + 1))
            return sup("dof-3 matrix: rows are not the 24 dofs of a hexahedron in DMDA order");
        const long el = (long)i + (long)ex * (j + (long)ey * k);
        if (A->E.empty()) A->E.assign((size_t)ex * ey * ez, 0.0);
        if (A->ref.empty()) A->ref.assign(y, y + 576);
        const double s = y[0] / A->ref[0];
        for (int q : {0, 1, 25, 300, 575})
            if (fabs(y[q] - s * A->ref[q]) > 1e-12 * fabs(s) * (fabs(A->ref[0]) + fabs(A->ref[q])))
                return sup("dof-3 matrix: element blocks are not multiples of one element matrix");
        if (getenv("TP_SHIM_VERIFY")) {  // the whole block and all 24 indices, not samples
            for (int q = 0; q < 576; q++)
                if (fabs(y[q] - s * A->ref[q]) > 1e-12 * fabs(s) * (fabs(A->ref[0]) + fabs(A->ref[q])))
                    return sup("dof-3 matrix (verify): an element block is not a multiple of the first one");
            const PetscInt dz = d->M * d->N;
            const PetscInt cell[8] = {n0, n0 + 1, n0 + 1 + d->M, n0 + d->M, n0 + dz, n0 + 1 + dz, n0 + 1 + d->M + dz, n0 + d->M + dz};
            for (int a = 0; a < 8; a++)
                for (int c = 0; c < 3; c++)
                    if (irow[3 * a + c] != 3 * cell[a] + c || icol[3 * a + c] != irow[3 * a + c])
                        return sup("dof-3 matrix (verify): rows/columns are not the hexahedron's dofs in the reference's corner order");
            A->nverified++;
        }
        A->E[(size_t)el] += s;
        A->ncalls++;
        A->assembled_since_setup = true;
        return 0;
    }
    case K_HELM:  // PDEFilt::MatAssemble, PDEFilter.cc:257-260: the same KF for every element
        if (nrow != 8 || ncol != 8 || addv != ADD_VALUES) return sup("dof-1 node matrix: 8x8 ADD_VALUES element blocks only");
        if (A->ref.empty()) A->ref.assign(y, y + 64);
        else if (memcmp(A->ref.data(), y, sizeof(double) * 64) != 0) return sup("dof-1 node matrix: variable coefficients");
        A->ncalls++;
        return 0;
    case K_TMAT:  // :262: T(8 nodes, element) = TF = 1/8
        if (nrow != 8 || ncol != 1 || addv != ADD_VALUES) return sup("transfer matrix: 8x1 ADD_VALUES blocks only");
        for (int q = 0; q < 8; q++)
            if (y[q] != 0.125) return sup("transfer matrix: entries other than 1/8");
        A->ncalls++;
        return 0;
    case K_CONE:  // Filter::SetUp, Filter.cc:417-433: H(row, col) = R - dist, INSERT_VALUES; H(row, row) = R
        if (nrow != 1 || ncol != 1 || addv != INSERT_VALUES) return sup("element matrix: 1x1 INSERT_VALUES entries only");
        if (irow[0] == icol[0]) {
            if (A->coneR == 0.0) A->coneR = y[0];
            else if (A->coneR != y[0]) return sup("element matrix: varying diagonal (not a cone filter of one radius)");
        } else if (A->coneR != 0.0 && !(y[0] > 0.0 && y[0] < A->coneR)) {
            return sup("element matrix: off-diagonal weight outside (0, R)");
        }
        if (getenv("TP_SHIM_VERIFY") && job_size() == 1) {  // keep the caller's matrix to check the device filter against it
            A->hrow.push_back((int)irow[0]);
            A->hcol.push_back((int)icol[0]);
            A->hval.push_back(y[0]);
        }
        A->ncalls++;
        return 0;
    default:
        return sup("MatSetValuesLocal on this matrix");
    }
}
PetscErrorCode MatAssemblyBegin(Mat, MatAssemblyType) { return 0; }
PetscErrorCode MatAssemblyEnd(Mat A, MatAssemblyType) {
    if (A->kind == K_CONE && !A->f) {
        if (A->coneR <= 0.0) return sup("element matrix without diagonal entries");
        int rc = ensure_grid();
        if (!rc) rc = tp_filter_create(&A->f, mesh.g, 1, A->coneR, nullptr);
        if (!rc && !A->hval.empty()) {
            // TP_SHIM_VERIFY=1: only the radius was taken from the caller's entries -- check that the device filter IS the
            // matrix the caller assembled (Filter.cc:417-433: the reference's own distances and weights): H x for a
            // pseudo-random x, entry by entry on the host, against tp_filter_mult_h
            const long n = A->n_rows;
            std::vector<double> x((size_t)n), yh((size_t)n, 0.0), yd((size_t)n);
            uint64_t st = 0x9E3779B97F4A7C15ULL;
            for (long i = 0; i < n; i++) {
                st = st * 6364136223846793005ULL + 1442695040888963407ULL;
                x[(size_t)i] = (double)(st >> 11) / 9007199254740992.0;
            }
            for (size_t e = 0; e < A->hval.size(); e++) yh[(size_t)A->hrow[e]] += A->hval[e] * x[(size_t)A->hcol[e]];
            double *dx = nullptr, *dy = nullptr;
            rc = tp_malloc((void **)&dx, sizeof(double) * (size_t)n) || tp_malloc((void **)&dy, sizeof(double) * (size_t)n);
            if (!rc) rc = tp_memcpy_h2d(dx, x.data(), sizeof(double) * (size_t)n);
            if (!rc) rc = tp_filter_mult_h(A->f, dx, dy);
            if (!rc) rc = tp_sync(mesh.g);
            if (!rc) rc = tp_memcpy_d2h(yd.data(), dy, sizeof(double) * (size_t)n);
            if (dx) tp_free(dx);
            if (dy) tp_free(dy);
            if (rc) return rc;
            double dev = 0.0, scale = 0.0;
            for (long i = 0; i < n; i++) {
                dev = fmax(dev, fabs(yh[(size_t)i] - yd[(size_t)i]));
                scale = fmax(scale, fabs(yh[(size_t)i]));
            }
            printf("[petsc-compat] verified the cone filter against the %zu inserted entries: max |H x - device| / max |H x| = %.3e\n",
                   A->hval.size(), dev / scale);
            if (!(dev <= 1e-12 * scale)) return sup("element matrix: the inserted entries are not the cone filter of their diagonal's radius on this mesh");
            A->hrow.clear();
            A->hcol.clear();
            A->hval.clear();
            A->hrow.shrink_to_fit();
            A->hcol.shrink_to_fit();
            A->hval.shrink_to_fit();
        }
        return rc;
    }
    return 0;
}
PetscErrorCode MatDiagonalScale(Mat A, Vec l, Vec r) {  // K = N K N, LinearElasticity.cc:533
    if (A->kind != K_ELAST || l != r || !l || l->n != A->n_rows) return sup("MatDiagonalScale: (K, N, N) on the stiffness matrix only");
    if (!A->Nvec) {
        int rc = VecDuplicate(l, &A->Nvec);
        if (rc) return rc;
    }
    A->have_bc = true;
    A->assembled_since_setup = true;
    return VecCopy(l, A->Nvec);
}
PetscErrorCode MatDiagonalSet(Mat A, Vec D, InsertMode mode) {  // K += I - N, :534-538
    if (A->kind != K_ELAST || mode != ADD_VALUES || !A->have_bc) return sup("MatDiagonalSet: (K, I - N, ADD_VALUES) after MatDiagonalScale only");
    double sd = 0.0, sn = 0.0;
    int rc = VecSum(D, &sd);
    if (!rc) rc = VecSum(A->Nvec, &sn);
    if (rc) return rc;
    if (sd + sn != (double)A->Nvec->nglob) return sup("MatDiagonalSet: the vector is not I - N");
    return 0;
}
PetscErrorCode MatMult(Mat A, Vec x, Vec y) {
    if (!A || !x || !y || x->n != A->n_cols || y->n != A->n_rows) return PETSC_ERR_ARG_WRONG;
    switch (A->kind) {
    case K_ELAST: {
        int rc = ensure_elasticity(A);
        if (rc) return rc;
        const double *px = binout(x);  // (the library refreshes the ghost planes of its input)
        return tp_elasticity_apply(A->e, px, bout(y));
    }
    case K_EXT_ELAST:
        {
        if (!A->ext_assembled) return PETSC_ERR_ORDER;
        const double *px = binout(x);
        return tp_elasticity_apply(A->e, px, bout(y));
    }
    case K_CONE:
        {
        if (!A->f) return PETSC_ERR_ORDER;
        const double *px = din(x);
        return tp_filter_mult_h(A->f, px, dout(y));
    }
    case K_EXT_FILTER:
        if (A->coneR < 0.0) {  // PDE filter as one operator
            const double *px = din(x);
            double *py = dout(y);
            return tp_filter_project(A->f, px, py, py, 0, 0.0, 0.0) ? PETSC_ERR_ARG_WRONG : 0;
        }
        {
            const double *px = din(x);
            return tp_filter_mult_h(A->f, px, dout(y));
        }
    case K_HELM: {
        int rc = ensure_pdefilter(A);
        if (rc) return rc;
        const double *px = binout(x);
        return tp_pdefilter_apply(A->f, px, bout(y));
    }
    case K_TMAT: {
        if (!g_last_helm) return PETSC_ERR_ORDER;
        int rc = ensure_pdefilter(g_last_helm);
        if (rc) return rc;
        const double *px = din(x);
        return tp_pdefilter_elem_to_node(g_last_helm->f, px, bout(y));
    }
    default:
        return sup("MatMult on this matrix");
    }
}
PetscErrorCode MatMultTranspose(Mat A, Vec x, Vec y) {
    if (!A || A->kind != K_TMAT) {
        if (A && (A->kind == K_ELAST || A->kind == K_HELM || A->kind == K_CONE || A->kind == K_EXT_ELAST)) return MatMult(A, x, y);  // symmetric
        return sup("MatMultTranspose on this matrix");
    }
    if (x->n != A->n_rows || y->n != A->n_cols || !g_last_helm) return PETSC_ERR_ARG_WRONG;
    int rc = ensure_pdefilter(g_last_helm);
    if (rc) return rc;
    const double *px = binout(x);  // its ghost planes are refreshed
    return tp_pdefilter_node_to_elem(g_last_helm->f, px, dout(y));
}
PetscErrorCode MatDestroy(Mat *A) {
    if (A && *A) {
        if (--(*A)->h.refct <= 0) {
            if ((*A)->e) tp_elasticity_destroy((*A)->e);
            if ((*A)->f) tp_filter_destroy((*A)->f);
            if ((*A)->dE) tp_free((*A)->dE);
            if ((*A)->Nvec) VecDestroy(&(*A)->Nvec);
            if (g_last_helm == *A) g_last_helm = nullptr;
            delete *A;
        }
        *A = nullptr;
    }
    return 0;
}

// =============================================================================================== KSP / PC
static KSP ksp_new(const char *prefix, bool sub) {
    KSP k = new _p_KSP();
    hdr_init(k->h, CLS_KSP, "ksp");
    k->type = KSPGMRES;  // PETSc's default
    k->prefix = prefix;
    k->rtol = 1e-5;
    k->atol = 1e-50;
    k->dtol = 1e5;
    k->maxits = 10000;
    k->restart = 30;
    k->nonzero_guess = false;
    k->from_options = false;
    k->A = nullptr;
    k->its = 0;
    k->rnorm = 0.0;
    k->is_sub = sub;
    k->pc = new _p_PC();
    hdr_init(k->pc->h, CLS_PC, "pc");
    k->pc->type = sub ? PCSOR : "ilu";  // PETSc's defaults (level smoothers: SOR)
    k->pc->nlevels = 0;
    k->pc->mgtype = PC_MG_MULTIPLICATIVE;
    k->pc->cycle = PC_MG_CYCLE_V;
    k->pc->galerkin = PC_MG_GALERKIN_NONE;
    k->pc->owner = k;
    return k;
}
PetscErrorCode KSPCreate(MPI_Comm, KSP *ksp) {
    *ksp = ksp_new("", false);
    return 0;
}
PetscErrorCode KSPSetType(KSP k, KSPType type) {
    k->type = type;
    return 0;
}
PetscErrorCode KSPGetType(KSP k, KSPType *type) {
    *type = k->type.c_str();
    return 0;
}
PetscErrorCode KSPGMRESSetRestart(KSP k, PetscInt restart) {
    k->restart = restart;
    return 0;
}
PetscErrorCode KSPSetTolerances(KSP k, PetscReal rtol, PetscReal abstol, PetscReal dtol, PetscInt maxits) {
    if (rtol != PETSC_DEFAULT) k->rtol = rtol;
    if (abstol != PETSC_DEFAULT) k->atol = abstol;
    if (dtol != PETSC_DEFAULT) k->dtol = dtol;
    if (maxits != PETSC_DEFAULT) k->maxits = maxits;
    return 0;
}
PetscErrorCode KSPGetTolerances(KSP k, PetscReal *rtol, PetscReal *abstol, PetscReal *dtol, PetscInt *maxits) {
    if (rtol) *rtol = k->rtol;
    if (abstol) *abstol = k->atol;
    if (dtol) *dtol = k->dtol;
    if (maxits) *maxits = k->maxits;
    return 0;
}
PetscErrorCode KSPSetInitialGuessNonzero(KSP k, PetscBool flg) {
    k->nonzero_guess = flg == PETSC_TRUE;
    return 0;
}
PetscErrorCode KSPSetOperators(KSP k, Mat A, Mat) {
    if (!A) return PETSC_ERR_ARG_WRONG;
    if (A != k->A) {
        A->h.refct++;
        if (k->A) MatDestroy(&k->A);
        k->A = A;
    }
    A->ksp = k;
    return 0;
}
PetscErrorCode KSPSetFromOptions(KSP k) {
    k->from_options = true;
    ksp_apply_options(k, {k->prefix});
    return 0;
}
// (extension, include/petsc_shim.h) what a configured KSP resolves to on the MI355X path, without touching the device:
// the tp_solver_opts the library would be created with, or PETSC_ERR_SUP
PetscErrorCode KSPCompatResolve(KSP k, tp_solver_opts *o) {
    if (!k || !o) return PETSC_ERR_ARG_WRONG;
    return resolve(k, o);
}
PetscErrorCode KSPSetUp(KSP k) {
    if (!k->A) return PETSC_ERR_ORDER;
    if (k->A->kind == K_ELAST) return ensure_elasticity(k->A);
    if (k->A->kind == K_HELM) return ensure_pdefilter(k->A);
    if (k->A->kind == K_EXT_ELAST) return k->A->ext_assembled ? 0 : PETSC_ERR_ORDER;
    return sup("KSP on this matrix");
}
PetscErrorCode KSPSolve(KSP k, Vec b, Vec x) {
    int rc = KSPSetUp(k);
    if (rc) return rc;
    Mat A = k->A;
    if (b->n != A->n_rows || x->n != A->n_rows) return PETSC_ERR_ARG_WRONG;
    if (!k->nonzero_guess) {
        rc = VecSet(x, 0.0);
        if (rc) return rc;
    }
    if (A->kind == K_HELM) {
        {
            const double *pb = bin(b);
            rc = tp_pdefilter_solve(A->f, pb, binout(x));
        }
        if (!rc) rc = tp_filter_last_pde_its(A->f, &k->its, &k->rnorm);
        return rc;
    }
    rc = tp_elasticity_set_tolerances(A->e, k->rtol, k->atol, k->dtol, k->maxits);
    if (rc) return rc;
    double bn = 0.0;
    const double *pb = bin(b);
    return tp_elasticity_solve(A->e, pb, binout(x), &k->its, &k->rnorm, &bn, nullptr, 0);
}
PetscErrorCode KSPGetIterationNumber(KSP k, PetscInt *its) {
    *its = k->its;
    return 0;
}
PetscErrorCode KSPGetResidualNorm(KSP k, PetscReal *rnorm) {
    *rnorm = k->rnorm;
    return 0;
}
PetscErrorCode KSPGetPC(KSP k, PC *pc) {
    *pc = k->pc;
    return 0;
}
PetscErrorCode KSPDestroy(KSP *k) {
    if (k && *k) {
        KSP s = *k;
        for (KSP sub : s->pc->lev) KSPDestroy(&sub);
        for (Mat m : s->pc->interp) MatDestroy(&m);
        if (s->A) {
            if (s->A->ksp == s) s->A->ksp = nullptr;
            MatDestroy(&s->A);
        }
        delete s->pc;
        delete s;
        *k = nullptr;
    }
    return 0;
}
PetscErrorCode PCSetType(PC pc, PCType type) {
    pc->type = type;
    return 0;
}
PetscErrorCode PCGetType(PC pc, PCType *type) {
    *type = pc->type.c_str();
    return 0;
}
PetscErrorCode PCSetReusePreconditioner(PC, PetscBool) { return 0; }
PetscErrorCode PCMGSetLevels(PC pc, PetscInt levels, MPI_Comm *) {
    if (levels < 1 || levels > TP_MAX_LEVELS) return PETSC_ERR_ARG_OUTOFRANGE;
    for (KSP sub : pc->lev) KSPDestroy(&sub);
    pc->lev.clear();
    pc->nlevels = levels;
    for (PetscInt l = 0; l < levels; l++) {
        KSP s = ksp_new(l == 0 && levels > 1 ? "mg_coarse_" : "mg_levels_", true);
        s->type = l == 0 && levels > 1 ? "preonly" : KSPCHEBYSHEV;  // PETSc's PCMG defaults
        s->pc->type = l == 0 && levels > 1 ? "lu" : PCSOR;
        s->maxits = l == 0 && levels > 1 ? 1 : 2;
        pc->lev.push_back(s);
    }
    pc->interp.assign((size_t)levels, nullptr);
    return 0;
}
PetscErrorCode PCMGSetType(PC pc, PCMGType form) {
    pc->mgtype = form;
    return 0;
}
PetscErrorCode PCMGSetCycleType(PC pc, PCMGCycleType n) {
    pc->cycle = n;
    return 0;
}
PetscErrorCode PCMGSetGalerkin(PC pc, PCMGGalerkinType use) {
    pc->galerkin = use;
    return 0;
}
PetscErrorCode PCMGSetInterpolation(PC pc, PetscInt l, Mat mat) {
    if (l < 1 || l >= pc->nlevels || !mat) return PETSC_ERR_ARG_OUTOFRANGE;
    if (mat->kind != K_INTERP) return sup("PCMGSetInterpolation: only the matrices of DMCreateInterpolation (trilinear Q1)");
    mat->h.refct++;  // retained: the caller destroys its reference (LinearElasticity.cc:704-706)
    if (pc->interp[(size_t)l]) MatDestroy(&pc->interp[(size_t)l]);
    pc->interp[(size_t)l] = mat;
    return 0;
}
PetscErrorCode PCMGGetCoarseSolve(PC pc, KSP *ksp) {
    if (pc->lev.empty()) return PETSC_ERR_ORDER;
    *ksp = pc->lev[0];
    return 0;
}
PetscErrorCode PCMGGetSmoother(PC pc, PetscInt l, KSP *ksp) {
    if (l < 0 || l >= (PetscInt)pc->lev.size()) return PETSC_ERR_ARG_OUTOFRANGE;
    *ksp = pc->lev[(size_t)l];
    return 0;
}
PetscErrorCode KSPTopOptGetOptionString(KSP k, char buf[], size_t len) {
    if (!k->A || !k->A->e) return PETSC_ERR_ORDER;
    return tp_elasticity_petsc_options(k->A->e, buf, len) < 0 ? PETSC_ERR_ORDER : 0;
}

// =============================================================================================== extension calls
PetscErrorCode MatCreateTopOptElasticity(DM da, PetscScalar nu, PetscInt nlvls, Mat *K) {
    DMFull *d = F(da);
    if (!da || d->dof != 3 || !is_nodal(d)) return PETSC_ERR_ARG_WRONG;
    if (d->have_box && !mesh.g) {
        memcpy(mesh.box, d->box, sizeof(mesh.box));
        mesh.have_box = true;
    }
    int rc = ensure_grid();
    if (rc) return rc;
    tp_solver_opts o;
    tp_solver_default_opts(&o);
    o.nlvls = nlvls;
    o.nu = nu;
    const long n = 3L * d->M * d->N * zbox(d, 0).zm;
    Mat A = mat_new(K_EXT_ELAST, da, n, n, "topopt-elasticity");
    rc = tp_elasticity_create(&A->e, mesh.g, &o);
    if (rc) {
        delete A;
        return rc;
    }
    *K = A;
    return 0;
}
PetscErrorCode MatTopOptCantilever(Mat K, Vec N, Vec RHS) {
    if (!K || K->kind != K_EXT_ELAST || N->n != K->n_rows || RHS->n != K->n_rows) return PETSC_ERR_ARG_WRONG;
    K->have_bc = true;
    return tp_elasticity_cantilever(K->e, bout(N), bout(RHS));  // also registers N
}
PetscErrorCode MatTopOptSetDirichlet(Mat K, Vec N) {
    if (!K || K->kind != K_EXT_ELAST || N->n != K->n_rows) return PETSC_ERR_ARG_WRONG;
    K->have_bc = true;
    int rc = job_size() > 1 ? tp_grid_halo_nodes(mesh.g, binout(N), 3) : 0;
    return rc ? rc : tp_elasticity_set_bc(K->e, bin(N));
}
PetscErrorCode MatTopOptAssemble(Mat K, Vec xPhys, PetscScalar Emin, PetscScalar Emax, PetscScalar penal) {
    if (!K || K->kind != K_EXT_ELAST) return PETSC_ERR_ARG_WRONG;
    if (!K->have_bc) return PETSC_ERR_ORDER;
    K->ext_assembled = true;
    return tp_elasticity_assemble(K->e, din(xPhys), Emin, Emax, penal);
}
PetscErrorCode MatTopOptComplianceSensitivity(Mat K, Vec U, Vec xPhys, PetscScalar Emin, PetscScalar Emax,
                                              PetscScalar penal, PetscScalar volfrac, PetscScalar *fx, PetscScalar *gx,
                                              Vec dfdx, Vec dgdx) {
    if (!K || (K->kind != K_EXT_ELAST && K->kind != K_ELAST) || !K->e) return PETSC_ERR_ARG_WRONG;
    const double *pu = binout(U), *px = din(xPhys);
    return tp_elasticity_objective(K->e, pu, px, Emin, Emax, penal, volfrac, fx, gx, dfdx ? dout(dfdx) : nullptr,
                                   dgdx ? dout(dgdx) : nullptr);
}
PetscErrorCode MatCreateTopOptFilter(DM da, PetscInt filterType, PetscScalar R, Mat *H, Vec *Hs) {
    DMFull *d = F(da);
    if (!da || filterType < 0 || filterType > 2 || !is_nodal(d)) return sup("MatCreateTopOptFilter: types 0, 1, 2 on the node mesh");
    if (d->have_box && !mesh.g) {
        memcpy(mesh.box, d->box, sizeof(mesh.box));
        mesh.have_box = true;
    }
    int rc = ensure_grid();
    if (rc) return rc;
    const long nel = tp_grid_local_elems(mesh.g);
    Mat A = mat_new(K_EXT_FILTER, da, nel, nel, "topopt-filter");
    A->coneR = filterType == 2 ? -1.0 : R;
    rc = tp_filter_create(&A->f, mesh.g, filterType, R, nullptr);
    if (rc) {
        delete A;
        return rc;
    }
    if (Hs) {
        const long per = (long)(d->M - 1) * (d->N - 1);
        rc = vec_create_layout(nel, 0, nel, per * (d->P - 1), per * ((d->P - 1) / job_size()) * job_rank(), job_size() == 1, false, nullptr, Hs);
        if (!rc) rc = filterType == 2 ? VecSet(*Hs, 1.0) : tp_filter_get_hs(A->f, dout(*Hs));
    }
    *H = A;
    return rc;
}

}  // extern "C"

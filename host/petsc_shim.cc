// petsc_shim.cc -- implementation of include/petsc_shim.h on top of the C ABI of libtopopt_amd.so.
// Pure host code (g++): every operation is one or two tp_* calls; no HIP, no PETSc.
#include "../include/petsc_shim.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

enum { ERR_ARG = 62 /* PETSC_ERR_ARG_OUTOFRANGE-like */, ERR_SUP = 56, ERR_ORDER = 58 };

struct _p_DM {
    PetscInt M, N, P, dof, s;
    double box[6];
    bool have_box, nodal;
    tp_grid *g;
};
struct _p_Vec {
    long n;
    double *d;
    std::vector<double> host;
    tp_grid *g;
};
struct _p_Mat {
    int kind;  // 0: elasticity, 1: cone filter
    tp_elasticity *e;
    tp_filter *f;
    tp_grid *g;
    long n_rows;
    bool have_bc, assembled;
};
struct _p_PC {
    int dummy;
};
struct _p_KSP {
    Mat A;
    double rtol, atol, dtol;
    int maxits;
    bool nonzero_guess;
    int its;
    double rnorm;
    _p_PC pc;
};

static tp_grid *g_default = nullptr;  // the grid element-sized vectors run their kernels on (one process)

static int grid_of(DM da, tp_grid **out) {
    if (!da->g) {
        if (!da->nodal) return ERR_ORDER;
        tp_grid_opts o;
        memset(&o, 0, sizeof(o));
        o.nx = da->M;
        o.ny = da->N;
        o.nz = da->P;
        if (da->have_box) {
            o.hx = (da->box[1] - da->box[0]) / (da->M - 1);
            o.hy = (da->box[3] - da->box[2]) / (da->N - 1);
            o.hz = (da->box[5] - da->box[4]) / (da->P - 1);
        } else {
            o.hx = o.hy = o.hz = 1.0;
        }
        o.rank = 0;
        o.nranks = 1;
        o.device = 0;
        int rc = tp_grid_create(&da->g, &o);
        if (rc) return rc;
        if (!g_default) g_default = da->g;
    }
    *out = da->g;
    return 0;
}

extern "C" {

PetscErrorCode PetscInitialize(int *, char ***, const char[], const char[]) { return 0; }
PetscErrorCode PetscFinalize(void) { return 0; }

PetscErrorCode DMDACreate3d(MPI_Comm, DMBoundaryType, DMBoundaryType, DMBoundaryType, DMDAStencilType, PetscInt M,
                            PetscInt N, PetscInt P, PetscInt, PetscInt, PetscInt, PetscInt dof, PetscInt s,
                            const PetscInt[], const PetscInt[], const PetscInt[], DM *da) {
    if (!da || M < 1 || N < 1 || P < 1 || dof < 1) return ERR_ARG;
    DM d = new _p_DM();
    d->M = M;
    d->N = N;
    d->P = P;
    d->dof = dof;
    d->s = s;
    d->have_box = false;
    d->nodal = false;
    d->g = nullptr;
    *da = d;
    return 0;
}
PetscErrorCode DMSetFromOptions(DM) { return 0; }
PetscErrorCode DMSetUp(DM) { return 0; }
PetscErrorCode DMDASetUniformCoordinates(DM da, PetscReal x0, PetscReal x1, PetscReal y0, PetscReal y1, PetscReal z0,
                                         PetscReal z1) {
    if (!da || da->g) return ERR_ORDER;  // before the first object that needs the grid
    const double b[6] = {x0, x1, y0, y1, z0, z1};
    memcpy(da->box, b, sizeof(b));
    da->have_box = true;
    return 0;
}
PetscErrorCode DMDAGetInfo(DM da, PetscInt *dim, PetscInt *M, PetscInt *N, PetscInt *P, PetscInt *m, PetscInt *n,
                           PetscInt *p, PetscInt *dof, PetscInt *s, DMBoundaryType *bx, DMBoundaryType *by,
                           DMBoundaryType *bz, DMDAStencilType *st) {
    if (dim) *dim = 3;
    if (M) *M = da->M;
    if (N) *N = da->N;
    if (P) *P = da->P;
    if (m) *m = 1;
    if (n) *n = 1;
    if (p) *p = 1;
    if (dof) *dof = da->dof;
    if (s) *s = da->s;
    if (bx) *bx = DM_BOUNDARY_NONE;
    if (by) *by = DM_BOUNDARY_NONE;
    if (bz) *bz = DM_BOUNDARY_NONE;
    if (st) *st = DMDA_STENCIL_BOX;
    return 0;
}
static PetscErrorCode vec_create(tp_grid *g, long n, Vec *v) {
    Vec x = new _p_Vec();
    x->n = n;
    x->g = g;
    x->d = nullptr;
    int rc = tp_malloc((void **)&x->d, sizeof(double) * (size_t)n);
    if (rc) {
        delete x;
        return rc;
    }
    *v = x;
    return tp_vec_set(g, x->d, 0.0, n);
}
PetscErrorCode DMCreateGlobalVector(DM da, Vec *v) {
    tp_grid *g = da->g ? da->g : g_default;
    if (!g) {  // the first vector of a program: this DM becomes the node grid
        da->nodal = true;
        int rc = grid_of(da, &g);
        if (rc) return rc;
    }
    return vec_create(g, (long)da->dof * da->M * da->N * da->P, v);
}
PetscErrorCode DMCreateLocalVector(DM da, Vec *v) { return DMCreateGlobalVector(da, v); }  // one rank: no ghosts
PetscErrorCode DMGlobalToLocalBegin(DM, Vec g, InsertMode, Vec l) { return g == l ? 0 : VecCopy(g, l); }
PetscErrorCode DMGlobalToLocalEnd(DM, Vec, InsertMode, Vec) { return 0; }
PetscErrorCode DMDestroy(DM *da) {
    if (da && *da) {
        if ((*da)->g) {
            if (g_default == (*da)->g) g_default = nullptr;
            tp_grid_destroy((*da)->g);
        }
        delete *da;
        *da = nullptr;
    }
    return 0;
}

PetscErrorCode VecDuplicate(Vec v, Vec *nv) { return vec_create(v->g, v->n, nv); }
PetscErrorCode VecDestroy(Vec *v) {
    if (v && *v) {
        tp_free((*v)->d);
        delete *v;
        *v = nullptr;
    }
    return 0;
}
PetscErrorCode VecSet(Vec v, PetscScalar a) { return tp_vec_set(v->g, v->d, a, v->n); }
PetscErrorCode VecCopy(Vec x, Vec y) {
    if (x->n != y->n) return ERR_ARG;
    return tp_vec_axpby(y->g, y->d, 1.0, x->d, 0.0, y->n);
}
PetscErrorCode VecScale(Vec v, PetscScalar a) { return tp_vec_scale(v->g, v->d, a, v->n); }
PetscErrorCode VecAXPY(Vec y, PetscScalar a, Vec x) {
    if (x->n != y->n) return ERR_ARG;
    return tp_vec_axpby(y->g, y->d, a, x->d, 1.0, y->n);
}
PetscErrorCode VecPointwiseMult(Vec w, Vec x, Vec y) {
    if (w->n != x->n || w->n != y->n) return ERR_ARG;
    return tp_vec_pointwise(w->g, w->d, x->d, y->d, 0, w->n);
}
PetscErrorCode VecPointwiseDivide(Vec w, Vec x, Vec y) {
    if (w->n != x->n || w->n != y->n) return ERR_ARG;
    return tp_vec_pointwise(w->g, w->d, x->d, y->d, 1, w->n);
}
PetscErrorCode VecDot(Vec x, Vec y, PetscScalar *val) {
    if (x->n != y->n) return ERR_ARG;
    return tp_vec_dot(x->g, x->d, y->d, x->n, val);
}
PetscErrorCode VecNorm(Vec x, NormType type, PetscReal *val) {
    if (type != NORM_2) return ERR_SUP;
    double s = 0.0;
    int rc = tp_vec_dot(x->g, x->d, x->d, x->n, &s);
    *val = std::sqrt(s);
    return rc;
}
PetscErrorCode VecSum(Vec x, PetscScalar *sum) { return tp_vec_dot(x->g, x->d, nullptr, x->n, sum); }
PetscErrorCode VecGetSize(Vec x, PetscInt *n) {
    *n = (PetscInt)x->n;
    return 0;
}
PetscErrorCode VecGetLocalSize(Vec x, PetscInt *n) { return VecGetSize(x, n); }
PetscErrorCode VecGetArray(Vec x, PetscScalar **a) {
    x->host.resize((size_t)x->n);
    tp_sync(x->g);
    int rc = tp_memcpy_d2h(x->host.data(), x->d, sizeof(double) * (size_t)x->n);
    *a = x->host.data();
    return rc;
}
PetscErrorCode VecRestoreArray(Vec x, PetscScalar **a) {
    if (a) *a = nullptr;
    return tp_memcpy_h2d(x->d, x->host.data(), sizeof(double) * (size_t)x->n);
}
PetscErrorCode VecTopOptGetDevicePointer(Vec x, PetscScalar **d) {
    *d = x->d;
    return 0;
}

PetscErrorCode MatCreateTopOptElasticity(DM da, PetscScalar nu, PetscInt nlvls, Mat *K) {
    if (!da || da->dof != 3) return ERR_ARG;
    da->nodal = true;
    tp_grid *g;
    int rc = grid_of(da, &g);
    if (rc) return rc;
    tp_solver_opts o;
    tp_solver_default_opts(&o);
    o.nlvls = nlvls;
    o.nu = nu;
    Mat A = new _p_Mat();
    A->kind = 0;
    A->g = g;
    A->f = nullptr;
    A->have_bc = A->assembled = false;
    A->n_rows = 3 * tp_grid_local_nodes(g);
    rc = tp_elasticity_create(&A->e, g, &o);
    if (rc) {
        delete A;
        return rc;
    }
    *K = A;
    return 0;
}
PetscErrorCode MatTopOptCantilever(Mat K, Vec N, Vec RHS) {
    if (!K || K->kind != 0 || N->n != K->n_rows || RHS->n != K->n_rows) return ERR_ARG;
    K->have_bc = true;
    return tp_elasticity_cantilever(K->e, N->d, RHS->d);  // also registers N
}
PetscErrorCode MatTopOptSetDirichlet(Mat K, Vec N) {
    if (!K || K->kind != 0 || N->n != K->n_rows) return ERR_ARG;
    K->have_bc = true;
    return tp_elasticity_set_bc(K->e, N->d);
}
PetscErrorCode MatTopOptAssemble(Mat K, Vec xPhys, PetscScalar Emin, PetscScalar Emax, PetscScalar penal) {
    if (!K || K->kind != 0) return ERR_ARG;
    if (!K->have_bc) return ERR_ORDER;
    K->assembled = true;
    return tp_elasticity_assemble(K->e, xPhys->d, Emin, Emax, penal);
}
PetscErrorCode MatTopOptComplianceSensitivity(Mat K, Vec U, Vec xPhys, PetscScalar Emin, PetscScalar Emax,
                                              PetscScalar penal, PetscScalar volfrac, PetscScalar *fx, PetscScalar *gx,
                                              Vec dfdx, Vec dgdx) {
    if (!K || K->kind != 0) return ERR_ARG;
    return tp_elasticity_objective(K->e, U->d, xPhys->d, Emin, Emax, penal, volfrac, fx, gx, dfdx ? dfdx->d : nullptr,
                                   dgdx ? dgdx->d : nullptr);
}
PetscErrorCode MatCreateTopOptFilter(DM da, PetscInt filterType, PetscScalar R, Mat *H, Vec *Hs) {
    if (!da || filterType < 0 || filterType > 1) return ERR_SUP;
    da->nodal = true;
    tp_grid *g;
    int rc = grid_of(da, &g);
    if (rc) return rc;
    Mat A = new _p_Mat();
    A->kind = 1;
    A->g = g;
    A->e = nullptr;
    A->n_rows = tp_grid_local_elems(g);
    A->have_bc = A->assembled = true;
    rc = tp_filter_create(&A->f, g, filterType, R, nullptr);
    if (rc) {
        delete A;
        return rc;
    }
    if (Hs) {
        rc = vec_create(g, A->n_rows, Hs);
        if (!rc) rc = tp_filter_get_hs(A->f, (*Hs)->d);
    }
    *H = A;
    return rc;
}
PetscErrorCode MatMult(Mat A, Vec x, Vec y) {
    if (!A || x->n != A->n_rows || y->n != A->n_rows) return ERR_ARG;
    if (A->kind == 0) return A->assembled ? tp_elasticity_apply(A->e, x->d, y->d) : ERR_ORDER;
    return tp_filter_mult_h(A->f, x->d, y->d);
}
PetscErrorCode MatDestroy(Mat *A) {
    if (A && *A) {
        if ((*A)->e) tp_elasticity_destroy((*A)->e);
        if ((*A)->f) tp_filter_destroy((*A)->f);
        delete *A;
        *A = nullptr;
    }
    return 0;
}

PetscErrorCode KSPCreate(MPI_Comm, KSP *ksp) {
    KSP k = new _p_KSP();
    k->A = nullptr;
    k->rtol = 1e-5;  // PETSc defaults
    k->atol = 1e-50;
    k->dtol = 1e5;
    k->maxits = 10000;
    k->nonzero_guess = false;
    k->its = 0;
    k->rnorm = 0.0;
    *ksp = k;
    return 0;
}
PetscErrorCode KSPSetType(KSP, KSPType type) { return strcmp(type, KSPCG) == 0 ? 0 : ERR_SUP; }
PetscErrorCode KSPSetTolerances(KSP k, PetscReal rtol, PetscReal abstol, PetscReal dtol, PetscInt maxits) {
    if (rtol != PETSC_DEFAULT) k->rtol = rtol;
    if (abstol != PETSC_DEFAULT) k->atol = abstol;
    if (dtol != PETSC_DEFAULT) k->dtol = dtol;
    if (maxits != PETSC_DEFAULT) k->maxits = maxits;
    return 0;
}
PetscErrorCode KSPSetInitialGuessNonzero(KSP k, PetscBool flg) {
    k->nonzero_guess = flg == PETSC_TRUE;
    return 0;
}
PetscErrorCode KSPSetOperators(KSP k, Mat A, Mat) {
    if (!A || A->kind != 0) return ERR_SUP;
    k->A = A;
    return 0;
}
PetscErrorCode KSPSetFromOptions(KSP) { return 0; }
PetscErrorCode KSPSetUp(KSP k) { return k->A && k->A->assembled ? 0 : ERR_ORDER; }  // Galerkin operators exist already
PetscErrorCode KSPSolve(KSP k, Vec b, Vec x) {
    if (!k->A || !k->A->assembled) return ERR_ORDER;
    if (b->n != k->A->n_rows || x->n != k->A->n_rows) return ERR_ARG;
    int rc = tp_elasticity_set_tolerances(k->A->e, k->rtol, k->atol, k->dtol, k->maxits);
    if (rc) return rc;
    if (!k->nonzero_guess) {
        rc = VecSet(x, 0.0);
        if (rc) return rc;
    }
    double bn = 0.0;
    return tp_elasticity_solve(k->A->e, b->d, x->d, &k->its, &k->rnorm, &bn, nullptr, 0);
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
    *pc = &k->pc;
    return 0;
}
PetscErrorCode KSPDestroy(KSP *k) {
    if (k && *k) {
        delete *k;
        *k = nullptr;
    }
    return 0;
}
PetscErrorCode PCSetType(PC, PCType type) { return strcmp(type, PCMG) == 0 ? 0 : ERR_SUP; }

}  // extern "C"

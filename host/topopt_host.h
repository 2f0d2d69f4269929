// topopt_host.h -- C++ host-side mirror of the reference's hot-path classes on top of the
// C ABI (include/topopt_amd.h).  Same class names, method names and argument meaning as
// LinearElasticity.h:21-109, Filter.h:34-92 and MMA.h:29-140; `Vec` is a device-resident
// vector with the small part of the PETSc Vec interface the driver (main.cc) uses.
// No PETSc, no torch: this is what a C++ caller of the library looks like.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/topopt_amd.h"

typedef int PetscErrorCode;
typedef double PetscScalar;
typedef int PetscInt;
#define CHKERRQ(ierr)                                                      \
    do {                                                                   \
        if (ierr) {                                                        \
            fprintf(stderr, "error %d at %s:%d\n", ierr, __FILE__, __LINE__); \
            return ierr;                                                   \
        }                                                                  \
    } while (0)

// ---- Vec: device storage + on-demand host mirror (VecGetArray / VecRestoreArray) ----------
struct Vec_ {
    double *d = nullptr;
    long n = 0;
    std::vector<double> h;
    tp_grid *g = nullptr;
};
typedef Vec_ *Vec;
inline PetscErrorCode VecCreate(tp_grid *g, long n, Vec *v) {
    *v = new Vec_();
    (*v)->n = n;
    (*v)->g = g;
    int e = tp_malloc((void **)&(*v)->d, sizeof(double) * (size_t)n);
    if (e) return e;
    return tp_vec_set(g, (*v)->d, 0.0, n);
}
inline PetscErrorCode VecDuplicate(Vec a, Vec *b) { return VecCreate(a->g, a->n, b); }
inline PetscErrorCode VecDestroy(Vec *v) {
    if (*v) {
        tp_free((*v)->d);
        delete *v;
        *v = nullptr;
    }
    return 0;
}
inline PetscErrorCode VecSet(Vec v, double a) { return tp_vec_set(v->g, v->d, a, v->n); }
inline PetscErrorCode VecScale(Vec v, double a) { return tp_vec_scale(v->g, v->d, a, v->n); }
inline PetscErrorCode VecGetArray(Vec v, double **p) {
    v->h.resize((size_t)v->n);
    tp_sync(v->g);
    int e = tp_memcpy_d2h(v->h.data(), v->d, sizeof(double) * (size_t)v->n);
    *p = v->h.data();
    return e;
}
inline PetscErrorCode VecRestoreArray(Vec v, double **p) {
    *p = nullptr;
    return tp_memcpy_h2d(v->d, v->h.data(), sizeof(double) * (size_t)v->n);
}

// ---- LinearElasticity (LinearElasticity.h:21-109) ---------------------------------------
class LinearElasticity {
  public:
    // smooth_its / coarse_its > 0: Chebyshev steps per smoothing sweep / of the coarse solve (-mg_levels_ksp_max_it,
    // -mg_coarse_ksp_max_it; the defaults are the reference's 4 and 30, DESIGN 4.5 has the measured optimum)
    LinearElasticity(tp_grid *grid, PetscInt nlvls_, PetscScalar nu_, PetscInt smooth_its = 0, PetscInt coarse_its = 0) : g(grid) {
        tp_solver_opts o;
        if (tp_abi_version() != TP_ABI_VERSION || tp_solver_opts_size() != sizeof(tp_solver_opts)) {  // header / library mismatch
            err = TP_ERR_STATE;
            return;
        }
        tp_solver_default_opts(&o);
        o.nlvls = nlvls_;
        o.nu = nu_;
        if (smooth_its > 0) o.nsmooth = smooth_its;
        if (coarse_its > 0) o.ncoarse = coarse_its;
        err = tp_elasticity_create(&e, g, &o);
        const long nn = 3 * tp_grid_local_nodes(g);
        VecCreate(g, nn, &U);
        VecCreate(g, nn, &RHS);
        VecCreate(g, nn, &N);
        if (!err) err = tp_elasticity_cantilever(e, N->d, RHS->d);  // SetUpLoadAndBC, LinearElasticity.cc:46-180
    }
    ~LinearElasticity() {
        VecDestroy(&U);
        VecDestroy(&RHS);
        VecDestroy(&N);
        tp_elasticity_destroy(e);
    }
    // LinearElasticity.cc:182-223
    PetscErrorCode SolveState(Vec xPhys, PetscScalar Emin, PetscScalar Emax, PetscScalar penal) {
        PetscErrorCode ierr = tp_elasticity_assemble(e, xPhys->d, Emin, Emax, penal);
        CHKERRQ(ierr);
        double rnorm, bnorm;
        ierr = tp_elasticity_solve(e, RHS->d, U->d, &niter, &rnorm, &bnorm, nullptr, 0);
        CHKERRQ(ierr);
        rerr = rnorm / bnorm;
        return 0;
    }
    // LinearElasticity.cc:363-445
    PetscErrorCode ComputeObjectiveConstraintsSensitivities(PetscScalar *fx, PetscScalar *gx, Vec dfdx, Vec dgdx,
                                                            Vec xPhys, PetscScalar Emin, PetscScalar Emax,
                                                            PetscScalar penal, PetscScalar volfrac) {
        PetscErrorCode ierr = SolveState(xPhys, Emin, Emax, penal);
        CHKERRQ(ierr);
        return tp_elasticity_objective(e, U->d, xPhys->d, Emin, Emax, penal, volfrac, fx, gx, dfdx->d, dgdx->d);
    }
    // LinearElasticity.cc:225-297 and :299-361: the split pair
    PetscErrorCode ComputeObjectiveConstraints(PetscScalar *fx, PetscScalar *gx, Vec xPhys, PetscScalar Emin, PetscScalar Emax,
                                               PetscScalar penal, PetscScalar volfrac) {
        PetscErrorCode ierr = SolveState(xPhys, Emin, Emax, penal);
        CHKERRQ(ierr);
        return tp_elasticity_objective_only(e, U->d, xPhys->d, Emin, Emax, penal, volfrac, fx, gx);
    }
    PetscErrorCode ComputeSensitivities(Vec dfdx, Vec dgdx, Vec xPhys, PetscScalar Emin, PetscScalar Emax, PetscScalar penal,
                                        PetscScalar /*volfrac*/) {
        return tp_elasticity_sensitivities(e, U->d, xPhys->d, Emin, Emax, penal, dfdx->d, dgdx->d);
    }
    Vec GetStateField() { return U; }
    PetscInt niter = 0;
    PetscScalar rerr = 0.0;
    PetscErrorCode err = 0;

  private:
    tp_grid *g;
    tp_elasticity *e = nullptr;
    Vec U = nullptr, RHS = nullptr, N = nullptr;
};

// ---- Filter (Filter.h:34-92) ---------------------------------------------------------------
class Filter {
  public:
    Filter(tp_grid *grid, PetscInt filterT, PetscScalar Rin) { err = tp_filter_create(&f, grid, filterT, Rin, nullptr); }
    ~Filter() { tp_filter_destroy(f); }
    PetscErrorCode FilterProject(Vec x, Vec xTilde, Vec xPhys, bool projectionFilter, PetscScalar beta, PetscScalar eta) {
        return tp_filter_project(f, x->d, xTilde->d, xPhys->d, projectionFilter, beta, eta);
    }
    PetscErrorCode Gradients(Vec x, Vec xTilde, Vec dfdx, PetscInt m, Vec *dgdx, bool projectionFilter, PetscScalar beta,
                             PetscScalar eta) {
        std::vector<double *> p((size_t)m);
        for (int i = 0; i < m; i++) p[(size_t)i] = dgdx[i]->d;
        return tp_filter_gradients(f, x->d, xTilde->d, dfdx->d, m, p.data(), projectionFilter, beta, eta);
    }
    PetscScalar GetMND(Vec x) {
        double v = 0.0;
        tp_filter_mnd(f, x->d, &v);
        return v;
    }
    PetscErrorCode err = 0;

  private:
    tp_filter *f = nullptr;
};

// ---- MMA (MMA.h:29-140) ----------------------------------------------------------------------
class MMA {
  public:
    MMA(tp_grid *grid, PetscInt n_global, PetscInt m_, Vec x) : m(m_) {
        err = tp_mma_create(&h, grid, x->n, n_global, m_, x->d);
    }
    ~MMA() { tp_mma_destroy(h); }
    PetscErrorCode SetOuterMovelimit(PetscScalar Xmin, PetscScalar Xmax, PetscScalar movlim, Vec x, Vec xmin, Vec xmax) {
        return tp_mma_set_outer_movelimit(h, Xmin, Xmax, movlim, x->d, xmin->d, xmax->d);
    }
    PetscErrorCode Update(Vec xval, Vec dfdx, PetscScalar *gx, Vec *dgdx, Vec xmin, Vec xmax) {
        std::vector<const double *> p((size_t)m);
        for (int i = 0; i < m; i++) p[(size_t)i] = dgdx[i]->d;
        return tp_mma_update(h, xval->d, dfdx->d, gx, p.data(), xmin->d, xmax->d, &inner);
    }
    PetscScalar DesignChange(Vec x, Vec xold) {
        double ch = 0.0;
        tp_mma_design_change(h, x->d, xold->d, &ch);
        return ch;
    }
    PetscErrorCode err = 0;
    int inner = 0;

  private:
    tp_mma *h = nullptr;
    PetscInt m;
};

// shim/ksp.cc -- PETSc-named surface of include/petsc_compat/petsc.h, part "ksp" (see shim/internal.h)
#include "internal.h"

extern "C" {

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

}  // extern "C"

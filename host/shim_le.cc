// shim_le.cc -- the reference's per-iteration code, in its own PETSc dialect, on the MI355X library.
//
// The bodies below follow LinearElasticity::SolveState (LinearElasticity.cc:182-223),
// ComputeObjectiveConstraintsSensitivities (:363-445), Filter::FilterProject (Filter.cc:60-117, filter type 1) and
// Filter::Gradients (:120-204) call by call; only what PETSc builds by assembly is replaced by the three
// MatTopOpt* extension calls of include/petsc_shim.h.  usage: shim_le nx ny nz nlvls [rmin_in_h]
#include <cstdio>
#include <cstdlib>

#include "../include/petsc_shim.h"

struct LinearElasticity {
    DM da_nodal;
    Mat K;
    KSP ksp;
    Vec U, RHS, N;
    PetscScalar nu;
    PetscInt nlvls;

    PetscErrorCode SetUp(PetscInt nx, PetscInt ny, PetscInt nz, const PetscScalar xc[6]) {
        PetscErrorCode ierr;
        // LinearElasticity.cc:60-105
        ierr = DMDACreate3d(PETSC_COMM_WORLD, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DMDA_STENCIL_BOX, nx, ny, nz,
                            PETSC_DECIDE, PETSC_DECIDE, PETSC_DECIDE, 3, 1, 0, 0, 0, &da_nodal);
        CHKERRQ(ierr);
        ierr = DMSetFromOptions(da_nodal); CHKERRQ(ierr);
        ierr = DMSetUp(da_nodal); CHKERRQ(ierr);
        ierr = DMDASetUniformCoordinates(da_nodal, xc[0], xc[1], xc[2], xc[3], xc[4], xc[5]); CHKERRQ(ierr);
        ierr = MatCreateTopOptElasticity(da_nodal, nu, nlvls, &K); CHKERRQ(ierr);   // DMCreateMatrix + MG hierarchy
        ierr = DMCreateGlobalVector(da_nodal, &U); CHKERRQ(ierr);
        ierr = VecDuplicate(U, &RHS); CHKERRQ(ierr);
        ierr = VecDuplicate(U, &N); CHKERRQ(ierr);
        ierr = MatTopOptCantilever(K, N, RHS); CHKERRQ(ierr);                         // :143-176
        // SetUpSolver, :617-650 (the PCG/GMG option string of DESIGN.md 1)
        ierr = KSPCreate(PETSC_COMM_WORLD, &ksp); CHKERRQ(ierr);
        ierr = KSPSetType(ksp, KSPCG); CHKERRQ(ierr);
        ierr = KSPSetTolerances(ksp, 1.0e-5, 1.0e-50, 1.0e3, 200); CHKERRQ(ierr);
        ierr = KSPSetInitialGuessNonzero(ksp, PETSC_TRUE); CHKERRQ(ierr);
        PC pc;
        ierr = KSPGetPC(ksp, &pc); CHKERRQ(ierr);
        ierr = PCSetType(pc, PCMG); CHKERRQ(ierr);
        return 0;
    }

    // LinearElasticity.cc:182-223
    PetscErrorCode SolveState(Vec xPhys, PetscScalar Emin, PetscScalar Emax, PetscScalar penal) {
        PetscErrorCode ierr;
        ierr = MatTopOptAssemble(K, xPhys, Emin, Emax, penal); CHKERRQ(ierr);        // AssembleStiffnessMatrix, :190
        ierr = KSPSetOperators(ksp, K, K); CHKERRQ(ierr);                            // :198
        ierr = KSPSetUp(ksp); CHKERRQ(ierr);                                         // :200
        ierr = KSPSolve(ksp, RHS, U); CHKERRQ(ierr);                                 // :204
        PetscInt niter;
        PetscScalar rnorm, RHSnorm;
        KSPGetIterationNumber(ksp, &niter);                                          // :210-217
        KSPGetResidualNorm(ksp, &rnorm);
        VecNorm(RHS, NORM_2, &RHSnorm);
        rnorm = rnorm / RHSnorm;
        printf("State solver:  iter: %i, rerr.: %e\n", niter, rnorm);
        return 0;
    }

    // LinearElasticity.cc:363-445
    PetscErrorCode ComputeObjectiveConstraintsSensitivities(PetscScalar *fx, PetscScalar *gx, Vec dfdx, Vec dgdx, Vec xPhys,
                                                            PetscScalar Emin, PetscScalar Emax, PetscScalar penal,
                                                            PetscScalar volfrac) {
        PetscErrorCode ierr;
        ierr = SolveState(xPhys, Emin, Emax, penal); CHKERRQ(ierr);                  // :376
        ierr = MatTopOptComplianceSensitivity(K, U, xPhys, Emin, Emax, penal, volfrac, fx, gx, dfdx, dgdx);  // :405-437
        CHKERRQ(ierr);
        return 0;
    }
};

struct Filter {
    Mat H;
    Vec Hs;
    PetscErrorCode SetUp(DM da_nodes, PetscScalar R) { return MatCreateTopOptFilter(da_nodes, 1, R, &H, &Hs); }  // Filter.cc:290-463
    // Filter.cc:60-117, filterType 1, no projection
    PetscErrorCode FilterProject(Vec x, Vec xTilde, Vec xPhys) {
        PetscErrorCode ierr;
        ierr = MatMult(H, x, xTilde); CHKERRQ(ierr);                                  // :68
        ierr = VecPointwiseDivide(xTilde, xTilde, Hs); CHKERRQ(ierr);                 // :70
        ierr = VecCopy(xTilde, xPhys); CHKERRQ(ierr);                                 // :113
        return 0;
    }
    // Filter.cc:180-192
    PetscErrorCode Gradients(Vec dfdx, Vec dgdx, Vec tmp) {
        PetscErrorCode ierr;
        ierr = VecPointwiseDivide(tmp, dfdx, Hs); CHKERRQ(ierr);
        ierr = MatMult(H, tmp, dfdx); CHKERRQ(ierr);
        ierr = VecPointwiseDivide(tmp, dgdx, Hs); CHKERRQ(ierr);
        ierr = MatMult(H, tmp, dgdx); CHKERRQ(ierr);
        return 0;
    }
};

int main(int argc, char **argv) {
    const PetscInt nx = argc > 1 ? atoi(argv[1]) : 33, ny = argc > 2 ? atoi(argv[2]) : 17, nz = argc > 3 ? atoi(argv[3]) : 17;
    const PetscInt nlvls = argc > 4 ? atoi(argv[4]) : 3;
    const double rmin_h = argc > 5 ? atof(argv[5]) : 2.56;
    PetscInitialize(&argc, &argv, 0, 0);
    const double h = 1.0 / (ny - 1);
    const PetscScalar xc[6] = {0.0, (nx - 1) * h, 0.0, 1.0, 0.0, (nz - 1) * h};
    LinearElasticity physics;
    physics.nu = 0.3;
    physics.nlvls = nlvls;
    PetscErrorCode ierr = physics.SetUp(nx, ny, nz, xc);
    if (ierr) { fprintf(stderr, "setup failed: %d\n", ierr); return 1; }
    // element fields (TopOpt.cc:362-381), all at volfrac
    DM da_elem;
    DMDACreate3d(PETSC_COMM_WORLD, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DMDA_STENCIL_BOX, nx - 1, ny - 1, nz - 1,
                 PETSC_DECIDE, PETSC_DECIDE, PETSC_DECIDE, 1, 0, 0, 0, 0, &da_elem);
    Vec x, xTilde, xPhys, dfdx, dgdx, tmp;
    DMCreateGlobalVector(da_elem, &x);
    VecDuplicate(x, &xTilde); VecDuplicate(x, &xPhys); VecDuplicate(x, &dfdx); VecDuplicate(x, &dgdx); VecDuplicate(x, &tmp);
    const PetscScalar volfrac = 0.12, Emin = 1e-9, Emax = 1.0, penal = 3.0;
    // a non-uniform design so that the filter does something: x = volfrac * (1 + 0.5 sin(i))
    PetscScalar *xp;
    PetscInt nel;
    VecGetArray(x, &xp);
    VecGetLocalSize(x, &nel);
    for (PetscInt i = 0; i < nel; i++) xp[i] = volfrac * (1.0 + 0.5 * ((i * 2654435761u) % 1000) / 1000.0);
    VecRestoreArray(x, &xp);
    Filter filter;
    ierr = filter.SetUp(physics.da_nodal, rmin_h * h);
    if (ierr) { fprintf(stderr, "filter setup failed: %d\n", ierr); return 1; }
    ierr = filter.FilterProject(x, xTilde, xPhys);                                   // main.cc:48
    PetscScalar fx, gx, s;
    for (int it = 0; it < 2 && !ierr; it++) {                                         // second pass: warm start
        ierr = physics.ComputeObjectiveConstraintsSensitivities(&fx, &gx, dfdx, dgdx, xPhys, Emin, Emax, penal, volfrac);
        if (ierr) break;
        ierr = filter.Gradients(dfdx, dgdx, tmp);
        VecSum(dfdx, &s);
        PetscScalar un;
        VecNorm(physics.U, NORM_2, &un);
        printf("fx: %.15e gx: %.15e sum(dfdx): %.15e |U|: %.15e\n", fx, gx, s, un);
    }
    if (ierr) { fprintf(stderr, "failed: %d\n", ierr); return 1; }
    VecDestroy(&x); VecDestroy(&xTilde); VecDestroy(&xPhys); VecDestroy(&dfdx); VecDestroy(&dgdx); VecDestroy(&tmp);
    VecDestroy(&physics.U); VecDestroy(&physics.RHS); VecDestroy(&physics.N); VecDestroy(&filter.Hs);
    MatDestroy(&filter.H); MatDestroy(&physics.K); KSPDestroy(&physics.ksp);
    DMDestroy(&da_elem); DMDestroy(&physics.da_nodal);
    PetscFinalize();
    return 0;
}

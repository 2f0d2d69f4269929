// ksp_probe -- which solver configuration does the compat layer make of the call sequence of the reference's two
// SetUpSolver methods (LinearElasticity.cc:617-746, PDEFilter.cc:275-378) plus the options on the command line?  Prints
// the resolved tp_solver_opts (or the PETSc error code).  No matrix, no vector: runs without a GPU.
//   ksp_probe le|pde nlvls [PETSc options]
#include <petsc.h>
#include <petsc_shim.h>

#include <cstdio>
#include <cstring>

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const bool pde = !strcmp(argv[1], "pde");
    const PetscInt nlvls = atoi(argv[2]);
    PetscInitialize(&argc, &argv, NULL, NULL);
    KSP ksp;
    PC pc;
    KSPCreate(PETSC_COMM_WORLD, &ksp);
    KSPSetType(ksp, KSPFGMRES);                              // :638 / PDEFilter.cc:276
    KSPGMRESSetRestart(ksp, pde ? 20 : 100);                 // :640 / :278
    if (pde) KSPSetTolerances(ksp, 1.0e-8, 1.0e-50, 1.0e3, 60);   // PDEFilter.cc:280-284
    else KSPSetTolerances(ksp, 1.0e-5, 1.0e-50, 1.0e5, 200);      // :621-625, :643
    KSPSetInitialGuessNonzero(ksp, PETSC_TRUE);
    KSPGetPC(ksp, &pc);
    PCSetType(pc, PCMG);
    KSPSetFromOptions(ksp);
    KSPGetPC(ksp, &pc);
    PetscBool is_mg = PETSC_FALSE;
    PetscObjectTypeCompare((PetscObject)pc, PCMG, &is_mg);
    if (is_mg) {
        PCMGSetLevels(pc, nlvls, NULL);
        PCMGSetType(pc, PC_MG_MULTIPLICATIVE);
        PCMGSetCycleType(pc, PC_MG_CYCLE_V);
        PCMGSetGalerkin(pc, PC_MG_GALERKIN_BOTH);
        KSP cksp;
        PC cpc;
        PCMGGetCoarseSolve(pc, &cksp);
        KSPSetType(cksp, KSPGMRES);
        KSPGMRESSetRestart(cksp, pde ? 10 : 30);
        KSPSetTolerances(cksp, 1.0e-8, 1.0e-50, pde ? 1e3 : 1e5, pde ? 10 : 30);
        KSPGetPC(cksp, &cpc);
        PCSetType(cpc, pde ? PCJACOBI : PCSOR);
        for (PetscInt k = 1; k < nlvls; k++) {
            KSP dksp;
            PC dpc;
            PCMGGetSmoother(pc, k, &dksp);
            KSPGetPC(dksp, &dpc);
            KSPSetType(dksp, KSPGMRES);
            KSPGMRESSetRestart(dksp, pde ? 1 : 4);
            KSPSetTolerances(dksp, PETSC_DEFAULT, PETSC_DEFAULT, PETSC_DEFAULT, pde ? 1 : 4);
            PCSetType(dpc, pde ? PCJACOBI : PCSOR);
        }
    }
    tp_solver_opts o;
    const PetscErrorCode ierr = KSPCompatResolve(ksp, &o);
    if (ierr) {
        printf("KSP_PROBE error %d\n", (int)ierr);
        KSPDestroy(&ksp);
        PetscFinalize();
        return 1;
    }
    printf("KSP_PROBE mode %d nlvls %d rtol %g atol %g dtol %g max_it %d nsmooth %d ncoarse %d restart %d smooth_pc %d coarse_pc %d "
           "coarse_restart %d coarse_rtol %g\n",
           o.ksp_mode, o.nlvls, o.rtol, o.atol, o.dtol, o.max_it, o.nsmooth, o.ncoarse, o.restart, o.smooth_pc, o.coarse_pc,
           o.coarse_restart, o.coarse_rtol);
    KSPDestroy(&ksp);
    PetscFinalize();
    return 0;
}

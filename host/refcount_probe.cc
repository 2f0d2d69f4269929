// refcount_probe -- object lifetimes at the PETSc-named boundary, without a GPU: the ownership pattern of the reference's
// SetUpSolver (LinearElasticity.cc:617-746: a coarsened DM hierarchy, interpolation matrices handed to PCMG and destroyed by
// the caller right away, the operator handed to the KSP and destroyed by the caller, borrowed sub-KSPs / PCs) replayed on the
// compat layer.  Meant to run under the sanitizer build (make asan): a reference dropped too early is a use-after-free, one
// never dropped a leak.  No vector is created: nothing touches the device.
//   refcount_probe nx ny nz nlvls
#include <petsc.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const PetscInt nx = atoi(argv[1]), ny = atoi(argv[2]), nz = atoi(argv[3]), nlvls = atoi(argv[4]);
    PetscInitialize(&argc, &argv, NULL, NULL);
    int rank = 0;
    MPI_Comm_rank(PETSC_COMM_WORLD, &rank);
    DM fine;
    if (DMDACreate3d(PETSC_COMM_WORLD, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DMDA_STENCIL_BOX, nx, ny, nz, PETSC_DECIDE,
                     PETSC_DECIDE, PETSC_DECIDE, 3, 1, 0, 0, 0, &fine))
        return 1;
    Mat K;
    if (DMCreateMatrix(fine, &K)) return 1;
    for (int round = 0; round < 2; round++) {  // twice: a solver torn down and built again on the same mesh and operator
        KSP ksp;
        PC pc;
        KSPCreate(PETSC_COMM_WORLD, &ksp);
        KSPSetType(ksp, KSPFGMRES);
        KSPGetPC(ksp, &pc);  // borrowed
        PCSetType(pc, PCMG);
        std::vector<DM> coarse((size_t)nlvls, nullptr), by_level((size_t)nlvls, nullptr);
        coarse[0] = fine;
        if (nlvls > 1 && DMCoarsenHierarchy(fine, nlvls - 1, &coarse[1])) return 1;
        for (PetscInt k = 0; k < nlvls; k++) by_level[(size_t)k] = coarse[(size_t)(nlvls - 1 - k)];  // PCMG counts from the coarsest
        PCMGSetLevels(pc, nlvls, NULL);
        PCMGSetGalerkin(pc, PC_MG_GALERKIN_BOTH);
        for (PetscInt k = 1; k < nlvls; k++) {
            Mat P;
            if (DMCreateInterpolation(by_level[(size_t)k - 1], by_level[(size_t)k], &P, NULL)) return 1;
            if (PCMGSetInterpolation(pc, k, P)) return 1;
            if (round == 1 && k == 1 && PCMGSetInterpolation(pc, k, P)) return 1;  // set twice: the first reference is released
            MatDestroy(&P);                                                         // PCMG keeps its own
        }
        for (PetscInt k = 1; k < nlvls; k++) DMDestroy(&coarse[(size_t)k]);  // the interpolations outlive their coarse meshes
        KSP sub;
        PC subpc;
        PCMGGetCoarseSolve(pc, &sub);  // borrowed
        KSPSetType(sub, KSPGMRES);
        KSPGetPC(sub, &subpc);
        PCSetType(subpc, PCSOR);
        for (PetscInt k = 1; k < nlvls; k++) {
            PCMGGetSmoother(pc, k, &sub);
            KSPSetType(sub, KSPGMRES);
            KSPGetPC(sub, &subpc);
            PCSetType(subpc, PCSOR);
        }
        KSPSetOperators(ksp, K, K);  // retained by the KSP
        KSPSetOperators(ksp, K, K);  // the same operator again: no second reference
        KSPDestroy(&ksp);
        if (ksp != NULL) return 3;
    }
    MatDestroy(&K);  // the caller's own reference, after both solvers are gone
    DMDestroy(&fine);
    printf("rank %d REFCOUNT_PROBE OK\n", rank);
    PetscFinalize();
    return 0;
}

// ref_driver.cc -- acceptance program of the PETSc-named boundary: the REFERENCE's own LinearElasticity / Filter /
// PDEFilt classes (compiled unchanged from /root/reference, in the build container only; nothing of them is stored in
// this repository) linked against libtopopt_petsc_shim.so, driven like main.cc:48-111 drives them for one design
// iteration.  This file is ours: it only calls the reference's public methods.
//   ref_on_shim ex ey ez filterType [petsc options...]
// (this acceptance program also looks at the class's load and Dirichlet vectors, which the reference keeps private)
#define private public
#include <LinearElasticity.h>
#undef private
#include <Filter.h>

#include <cstdint>

static double hash_u01(uint64_t idx, uint64_t seed) {  // the synthetic field of SURVEY.md 8(d) (csrc/common.h)
    uint64_t z = (idx + 1u) * 0x9E3779B97F4A7C15ULL + seed;
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const PetscInt ex = atoi(argv[1]), ey = atoi(argv[2]), ez = atoi(argv[3]), filterType = atoi(argv[4]);
    PetscInitialize(&argc, &argv, NULL, NULL);
    const PetscInt nx = ex + 1, ny = ey + 1, nz = ez + 1;
    const double h = 1.0 / ey, rmin = 2.56 * h;
    PetscErrorCode ierr;
    // TopOpt::SetUpMESH (TopOpt.cc:225-300): node mesh (dof 1, stencil 1) and the element mesh of the design field
    DM da_nodes, da_elem;
    ierr = DMDACreate3d(PETSC_COMM_WORLD, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DMDA_STENCIL_BOX, nx, ny, nz,
                        PETSC_DECIDE, PETSC_DECIDE, PETSC_DECIDE, 1, 1, 0, 0, 0, &da_nodes);
    CHKERRQ(ierr);
    DMSetFromOptions(da_nodes);
    DMSetUp(da_nodes);
    DMDASetUniformCoordinates(da_nodes, 0.0, ex * h, 0.0, ey * h, 0.0, ez * h);
    DMDASetElementType(da_nodes, DMDA_ELEMENT_Q1);
    PetscInt md, nd, pd;  // the element mesh lives on the process grid of the node mesh (TopOpt.cc:254-290)
    DMDAGetInfo(da_nodes, NULL, NULL, NULL, NULL, &md, &nd, &pd, NULL, NULL, NULL, NULL, NULL, NULL);
    ierr = DMDACreate3d(PETSC_COMM_WORLD, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DMDA_STENCIL_BOX, ex, ey, ez,
                        md, nd, pd, 1, 0, 0, 0, 0, &da_elem);
    CHKERRQ(ierr);
    DMSetUp(da_elem);
    Vec x, xTilde, xPhys, dfdx, dgdx;
    ierr = DMCreateGlobalVector(da_elem, &x);
    CHKERRQ(ierr);
    VecDuplicate(x, &xTilde);
    VecDuplicate(x, &xPhys);
    VecDuplicate(x, &dfdx);
    VecDuplicate(x, &dgdx);
    {
        PetscScalar *xp;
        VecGetArray(x, &xp);
        const double pi = 3.14159265358979323846;
        PetscInt xs, ys, zs, xm, ym, zm;  // this rank's elements
        DMDAGetCorners(da_elem, &xs, &ys, &zs, &xm, &ym, &zm);
        long at = 0;
        for (PetscInt k = zs; k < zs + zm; k++)
            for (PetscInt j = ys; j < ys + ym; j++)
                for (PetscInt i = xs; i < xs + xm; i++) {
                    const uint64_t gid = (uint64_t)i + (uint64_t)ex * ((uint64_t)j + (uint64_t)ey * (uint64_t)k);
                    double v = 0.12 + 0.4 * sin(7 * pi * (i + 0.5) * h) * sin(5 * pi * (j + 0.5) * h) * sin(3 * pi * (k + 0.5) * h) +
                               0.3 * (hash_u01(gid, 12345) - 0.5);
                    xp[at++] = v < 1e-3 ? 1e-3 : (v > 1.0 ? 1.0 : v);
                }
        VecRestoreArray(x, &xp);
    }
    LinearElasticity *physics = new LinearElasticity(da_nodes);        // main.cc:33
    Filter *filter = new Filter(da_nodes, x, filterType, rmin);         // main.cc:36
    const PetscScalar Emin = 1e-9, Emax = 1.0, penal = 3.0, volfrac = 0.12;
    ierr = filter->FilterProject(x, xTilde, xPhys, PETSC_FALSE, 0.1, 0.0);  // main.cc:48
    CHKERRQ(ierr);
    PetscScalar fx = 0, gx = 0;
    ierr = physics->ComputeObjectiveConstraintsSensitivities(&fx, &gx, dfdx, dgdx, xPhys, Emin, Emax, penal, volfrac);  // :62
    if (ierr) {
        printf("REF_ON_SHIM failed: %d\n", ierr);
        return 1;
    }
    ierr = filter->Gradients(x, xTilde, dfdx, 1, &dgdx, PETSC_FALSE, 0.1, 0.0);  // :76
    CHKERRQ(ierr);
    PetscScalar sdf = 0, sdg = 0, sxp = 0;
    PetscReal un = 0;
    VecSum(dfdx, &sdf);
    VecSum(dgdx, &sdg);
    VecSum(xPhys, &sxp);
    VecNorm(physics->GetStateField(), NORM_2, &un);
    if (const char *dump = getenv("REF_ON_SHIM_DUMP")) {  // xPhys, dfdx, dgdx (filtered, as main.cc:76 leaves them), U
        PetscViewer view;
        ierr = PetscViewerBinaryOpen(PETSC_COMM_WORLD, dump, FILE_MODE_WRITE, &view);
        CHKERRQ(ierr);
        VecView(xPhys, view);
        VecView(dfdx, view);
        VecView(dgdx, view);
        VecView(physics->GetStateField(), view);
        VecView(physics->N, view);    // what the reference's SetUpLoadAndBC (LinearElasticity.cc:46-180) set
        VecView(physics->RHS, view);  // (after AssembleStiffnessMatrix zeroed the loads on clamped dofs, :541)
        PetscViewerDestroy(&view);
    }
    PetscInt its = 0;
    PetscPrintf(PETSC_COMM_WORLD, "REF_ON_SHIM fx %.16e gx %.16e sum_dfdx %.16e sum_dgdx %.16e sum_xphys %.16e normU %.16e\n", fx, gx, sdf, sdg, sxp, un);
    delete filter;
    delete physics;
    VecDestroy(&x);
    VecDestroy(&xTilde);
    VecDestroy(&xPhys);
    VecDestroy(&dfdx);
    VecDestroy(&dgdx);
    DMDestroy(&da_elem);
    DMDestroy(&da_nodes);
    PetscFinalize();
    return 0;
}

// ref_mma_driver.cc -- acceptance program for the MMA row: the REFERENCE's own MMA class (MMA.cc, compiled unchanged
// from /root/reference in the build container; nothing of it is stored here) on the compat layer, driven with a
// synthetic smooth problem of m constraints for a few iterations; the design vector of every iteration goes to a PETSc
// binary file.  tests/test_mma.py feeds the same functions to the device MMA (tp_mma_*).  This file is ours.
//   ref_mma ex ey ez m iters out.bin
#include <MMA.h>
#include <petsc.h>

#include <cmath>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 7) return 2;
    const PetscInt ex = atoi(argv[1]), ey = atoi(argv[2]), ez = atoi(argv[3]), m = atoi(argv[4]), iters = atoi(argv[5]);
    PetscInitialize(&argc, &argv, NULL, NULL);
    PetscErrorCode ierr;
    DM nodes, elems;  // the node mesh defines the job's mesh; the design lives on its element mesh
    ierr = DMDACreate3d(PETSC_COMM_WORLD, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DMDA_STENCIL_BOX, ex + 1, ey + 1,
                        ez + 1, PETSC_DECIDE, PETSC_DECIDE, PETSC_DECIDE, 1, 1, 0, 0, 0, &nodes);
    CHKERRQ(ierr);
    DMDASetUniformCoordinates(nodes, 0.0, (double)ex / ey, 0.0, 1.0, 0.0, (double)ez / ey);
    PetscInt md, nd, pd;
    DMDAGetInfo(nodes, NULL, NULL, NULL, NULL, &md, &nd, &pd, NULL, NULL, NULL, NULL, NULL, NULL);
    ierr = DMDACreate3d(PETSC_COMM_WORLD, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DMDA_STENCIL_BOX, ex, ey, ez, md, nd,
                        pd, 1, 0, 0, 0, 0, &elems);
    CHKERRQ(ierr);
    Vec x, dfdx, xmin, xmax, xold;
    ierr = DMCreateGlobalVector(elems, &x);
    CHKERRQ(ierr);
    VecDuplicate(x, &dfdx);
    VecDuplicate(x, &xmin);
    VecDuplicate(x, &xmax);
    VecDuplicate(x, &xold);
    Vec *dgdx;
    VecDuplicateVecs(x, m, &dgdx);
    PetscInt n, nloc, zs;
    VecGetSize(x, &n);
    VecGetLocalSize(x, &nloc);
    DMDAGetCorners(elems, NULL, NULL, &zs, NULL, NULL, NULL);
    const long g0 = (long)zs * ex * ey;  // global index of this rank's first element
    VecSet(x, 0.3);
    VecSet(xold, 0.3);
    MMA *mma = new MMA(n, m, x);
    PetscViewer view;
    ierr = PetscViewerBinaryOpen(PETSC_COMM_WORLD, argv[6], FILE_MODE_WRITE, &view);
    CHKERRQ(ierr);
    std::vector<PetscScalar> gx((size_t)m);
    for (PetscInt k = 0; k < iters; k++) {
        PetscScalar *xp, *dfp;
        VecGetArray(x, &xp);
        VecGetArray(dfdx, &dfp);
        std::vector<double> gl((size_t)m, 0.0);
        for (PetscInt i = 0; i < nloc; i++) {  // f = sum_i a_i / (x_i + 0.1): df/dx_i = -a_i / (x_i + 0.1)^2
            const double a = 1.0 + 0.3 * sin(0.37 * (double)(g0 + i));
            dfp[i] = -a / ((xp[i] + 0.1) * (xp[i] + 0.1));
        }
        for (PetscInt j = 0; j < m; j++) {  // g_j = sum_i w_ji x_i / n - c_j
            PetscScalar *gp;
            VecGetArray(dgdx[j], &gp);
            for (PetscInt i = 0; i < nloc; i++) {
                const double w = 1.0 + 0.5 * cos(0.11 * (double)(g0 + i) * (double)(j + 1));
                gp[i] = w / (double)n;
                gl[(size_t)j] += w * xp[i] / (double)n;
            }
            VecRestoreArray(dgdx[j], &gp);
        }
        VecRestoreArray(x, &xp);
        VecRestoreArray(dfdx, &dfp);
        MPI_Allreduce(gl.data(), gx.data(), (int)m, MPI_DOUBLE, MPI_SUM, PETSC_COMM_WORLD);
        for (PetscInt j = 0; j < m; j++) gx[(size_t)j] -= 0.25 + 0.05 * (double)j;
        ierr = mma->SetOuterMovelimit(0.0, 1.0, 0.2, x, xmin, xmax);
        CHKERRQ(ierr);
        ierr = mma->Update(x, dfdx, gx.data(), dgdx, xmin, xmax);
        CHKERRQ(ierr);
        const PetscScalar ch = mma->DesignChange(x, xold);
        PetscPrintf(PETSC_COMM_WORLD, "REF_MMA it %d ch %.12e g0 %.12e\n", (int)(k + 1), ch, gx[0]);
        VecView(x, view);
    }
    PetscViewerDestroy(&view);
    delete mma;
    VecDestroyVecs(m, &dgdx);
    for (Vec *v : {&x, &dfdx, &xmin, &xmax, &xold}) VecDestroy(v);
    DMDestroy(&elems);
    DMDestroy(&nodes);
    PetscFinalize();
    return 0;
}

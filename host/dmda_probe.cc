// dmda_probe -- prints what the compat layer's DMDA tells each rank of a slab job about its part of the mesh (ownership,
// ghost ranges, local sizes), for a node mesh, the element mesh built on its ownership ranges the way TopOpt.cc:254-290
// and Filter.cc:337-364 do it, and a wide-stencil element mesh.  No vector is created: runs without a GPU.
//   slabrun -n R dmda_probe nx ny nz sw
#include <petsc.h>

#include <cstdio>
#include <cstdlib>
#include <initializer_list>

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const PetscInt nx = atoi(argv[1]), ny = atoi(argv[2]), nz = atoi(argv[3]), sw = atoi(argv[4]);
    PetscInitialize(&argc, &argv, NULL, NULL);
    int rank = 0, size = 1;
    MPI_Comm_rank(PETSC_COMM_WORLD, &rank);
    MPI_Comm_size(PETSC_COMM_WORLD, &size);
    DM nodes, elems;
    PetscErrorCode ierr = DMDACreate3d(PETSC_COMM_WORLD, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DMDA_STENCIL_BOX, nx, ny,
                                       nz, PETSC_DECIDE, PETSC_DECIDE, PETSC_DECIDE, 3, 1, 0, 0, 0, &nodes);
    if (ierr) return 1;
    PetscInt md, nd, pd;
    DMDAGetInfo(nodes, NULL, NULL, NULL, NULL, &md, &nd, &pd, NULL, NULL, NULL, NULL, NULL, NULL);
    const PetscInt *lx, *ly, *lz;
    DMDAGetOwnershipRanges(nodes, &lx, &ly, &lz);
    PetscInt *Lx = new PetscInt[md], *Ly = new PetscInt[nd], *Lz = new PetscInt[pd];
    for (int i = 0; i < md; i++) Lx[i] = lx[i] - (i == 0);
    for (int i = 0; i < nd; i++) Ly[i] = ly[i] - (i == 0);
    for (int i = 0; i < pd; i++) Lz[i] = lz[i] - (i == 0);
    ierr = DMDACreate3d(PETSC_COMM_WORLD, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DMDA_STENCIL_BOX, nx - 1, ny - 1,
                        nz - 1, md, nd, pd, 1, sw, Lx, Ly, Lz, &elems);
    if (ierr) return 1;
    for (DM dm : {nodes, elems}) {
        PetscInt xs, ys, zs, xm, ym, zm, gxs, gys, gzs, gxm, gym, gzm;
        DMDAGetCorners(dm, &xs, &ys, &zs, &xm, &ym, &zm);
        DMDAGetGhostCorners(dm, &gxs, &gys, &gzs, &gxm, &gym, &gzm);
        DMDALocalInfo info;
        DMDAGetLocalInfo(dm, &info);
        printf("rank %d of %d %s grid %d %d %d own %d %d %d + %d %d %d ghost %d %d %d + %d %d %d info %d %d %d %d sw %d\n", rank, size,
               dm == nodes ? "nodes" : "elems", md, nd, pd, xs, ys, zs, xm, ym, zm, gxs, gys, gzs, gxm, gym, gzm, info.zs, info.zm,
               info.gzs, info.gzm, info.sw);
    }
    delete[] Lx;
    delete[] Ly;
    delete[] Lz;
    DMDestroy(&elems);
    DMDestroy(&nodes);
    PetscFinalize();
    return 0;
}

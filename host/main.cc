// main.cc -- the reference's driver loop (main.cc:22-141) written against the C++ host mirror:
// construct -> initial filter -> { solve + sensitivities, scale, filter gradients, move limits, MMA,
// change, filter, MND, print }.  Command line: -nx -ny -nz (node counts, TopOpt.cc:154-160), -nlvls,
// -maxItr, -filter, -rmin, -volfrac, -penal, -mg_levels_ksp_max_it, -mg_coarse_ksp_max_it (Chebyshev steps of the
// V-cycle; e.g. -nlvls 5 -mg_levels_ksp_max_it 2 -mg_coarse_ksp_max_it 45).  Output/restart files are out
// of scope.
// One process per GPU: `host/slabrun -n N ./host/topopt ...` starts the z-slab ranks (slab_comm.h: shared-memory
// hooks, upgraded to the library's RCCL path where every rank has a GPU of its own).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>

#include "slab_comm.h"
#include "topopt_host.h"

static double opt_d(int argc, char **argv, const char *k, double d) {
    for (int i = 1; i + 1 < argc; i++)
        if (!strcmp(argv[i], k)) return atof(argv[i + 1]);
    return d;
}

int main(int argc, char **argv) {
    // TopOpt.cc:106-135 defaults
    const int nx = (int)opt_d(argc, argv, "-nx", 65), ny = (int)opt_d(argc, argv, "-ny", 33), nz = (int)opt_d(argc, argv, "-nz", 33);
    const int nlvls = (int)opt_d(argc, argv, "-nlvls", 4), maxItr = (int)opt_d(argc, argv, "-maxItr", 400);
    const int filterType = (int)opt_d(argc, argv, "-filter", 1), m = 1;
    const double xc[6] = {0.0, 2.0, 0.0, 1.0, 0.0, 1.0};
    const double nu = 0.3, volfrac = opt_d(argc, argv, "-volfrac", 0.12), rmin = opt_d(argc, argv, "-rmin", 0.08);
    const double penal = opt_d(argc, argv, "-penal", 3.0), Emin = 1.0e-9, Emax = 1.0, Xmin = 0.0, Xmax = 1.0, movlim = 0.2;
    const bool projectionFilter = false;
    double beta = 0.1, eta = 0.0;

    SlabComm sc;
    if (slab_comm_init(&sc, std::max(4L * 3 * nx * ny, 1L << 16))) return 1;
    const bool root = sc.rank == 0;
    tp_grid_opts go = {nx, ny, nz, (xc[1] - xc[0]) / (nx - 1), (xc[3] - xc[2]) / (ny - 1), (xc[5] - xc[4]) / (nz - 1),
                       sc.rank, sc.nranks, sc.device, nullptr, sc.nranks > 1 ? &sc.hooks : nullptr};
    tp_grid *grid = nullptr;
    PetscErrorCode ierr = tp_grid_create(&grid, &go);
    CHKERRQ(ierr);
    sc.grid = grid;
    slab_comm_try_rccl(&sc, grid);
    const long nel = tp_grid_local_elems(grid);
    if (root)
        printf("# nodes %d x %d x %d, %ld elements and %ld DOF on rank 0 of %d, nlvls %d, filter %d, rmin %g\n", nx, ny, nz, nel,
               3 * tp_grid_owned_nodes(grid), sc.nranks, nlvls, filterType, rmin);

    LinearElasticity *physics = new LinearElasticity(grid, nlvls, nu, (int)opt_d(argc, argv, "-mg_levels_ksp_max_it", 0),
                                                     (int)opt_d(argc, argv, "-mg_coarse_ksp_max_it", 0));
    CHKERRQ(physics->err);
    Filter *filter = new Filter(grid, filterType, rmin);
    CHKERRQ(filter->err);
    Vec x, xTilde, xPhys, dfdx, dgdx[1], xmin, xmax, xold;
    for (Vec *v : {&x, &xTilde, &xPhys, &dfdx, &dgdx[0], &xmin, &xmax, &xold}) VecCreate(grid, nel, v);
    for (Vec v : {x, xTilde, xPhys, xold}) VecSet(v, volfrac);  // TopOpt.cc:362-381
    MMA *mma = new MMA(grid, (PetscInt)nel, m, x);
    CHKERRQ(mma->err);

    ierr = filter->FilterProject(x, xTilde, xPhys, projectionFilter, beta, eta);  // main.cc:48
    CHKERRQ(ierr);
    double fx = 0.0, gx[1] = {0.0}, fscale = 1.0, ch = 1.0;
    int itr = 0;
    while (itr < maxItr && ch > 0.01) {  // main.cc:54
        itr++;
        auto t1 = std::chrono::steady_clock::now();
        ierr = physics->ComputeObjectiveConstraintsSensitivities(&fx, &gx[0], dfdx, dgdx[0], xPhys, Emin, Emax, penal,
                                                                 volfrac);
        CHKERRQ(ierr);
        if (itr == 1) fscale = 10.0 / fx;
        fx = fx * fscale;
        VecScale(dfdx, fscale);
        ierr = filter->Gradients(x, xTilde, dfdx, m, dgdx, projectionFilter, beta, eta);
        CHKERRQ(ierr);
        ierr = mma->SetOuterMovelimit(Xmin, Xmax, movlim, x, xmin, xmax);
        CHKERRQ(ierr);
        ierr = mma->Update(x, dfdx, gx, dgdx, xmin, xmax);
        CHKERRQ(ierr);
        ch = mma->DesignChange(x, xold);
        ierr = filter->FilterProject(x, xTilde, xPhys, projectionFilter, beta, eta);
        CHKERRQ(ierr);
        const double mnd = filter->GetMND(xPhys);
        tp_sync(grid);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        if (root) {
            printf("State solver:  iter: %i, rerr.: %e\n", physics->niter, physics->rerr);
            printf("It.: %i, True fx: %f, Scaled fx: %f, gx[0]: %f, ch.: %f, mnd.: %f, time: %f\n", itr, fx / fscale, fx,
                   gx[0], ch, mnd, dt);
        }
    }
    // a host-side look at the design through the Vec mirror (what MPIIO would dump)
    double *xp;
    VecGetArray(xPhys, &xp);
    double s = 0.0;
    for (long i = 0; i < nel; i++) s += xp[i];
    VecRestoreArray(xPhys, &xp);
    double tot[2] = {s, (double)nel};
    slab_detail::host_reduce(&sc, tot, 2, 0);
    if (root) printf("# final volume fraction %.6f\n", tot[0] / tot[1]);
    delete mma;
    delete filter;
    delete physics;
    for (Vec *v : {&x, &xTilde, &xPhys, &dfdx, &dgdx[0], &xmin, &xmax, &xold}) VecDestroy(v);
    tp_grid_destroy(grid);
    slab_comm_free(&sc);
    return 0;
}

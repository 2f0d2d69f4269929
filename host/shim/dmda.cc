// shim/dmda.cc -- PETSc-named surface of include/petsc_compat/petsc.h, part "dmda" (see shim/internal.h)
#include "internal.h"

extern "C" {

// =============================================================================================== DMDA
PetscErrorCode DMDACreate3d(MPI_Comm, DMBoundaryType, DMBoundaryType, DMBoundaryType, DMDAStencilType, PetscInt M,
                            PetscInt N, PetscInt P, PetscInt m, PetscInt n, PetscInt p, PetscInt dof, PetscInt s,
                            const PetscInt[], const PetscInt[], const PetscInt lz[], DM *da) {
    if (!da || M < 1 || N < 1 || P < 1 || dof < 1) return PETSC_ERR_ARG_OUTOFRANGE;
    // the process grid is 1 x 1 x R (z-slabs): PETSC_DECIDE resolves to it, anything else is refused
    const int R = job_size();
    if ((m != PETSC_DECIDE && m != 1) || (n != PETSC_DECIDE && n != 1) || (p != PETSC_DECIDE && p != R))
        return sup("DMDACreate3d: the ranks of the job form a 1 x 1 x R process grid (z-slabs)");
    int zkind = 0;
    if (R > 1) {
        if ((P - 1) % R == 0) zkind = 0;
        else if (P % R == 0) zkind = 1;
        else return sup("DMDACreate3d: the z extent does not split into equal slabs of elements over the ranks");
        const PetscInt e = zkind == 0 ? (P - 1) / R : P / R;
        if (lz)
            for (int q = 0; q < R; q++)
                if (lz[q] != e + ((zkind == 0 && q == 0) ? 1 : 0)) return sup("DMDACreate3d: lz[] is not the slab partition of the node mesh");
    }
    DMFull *d = new DMFull();
    d->zkind = zkind;
    for (int q = 0; q < R; q++) d->lzv.push_back(R == 1 ? P : (zkind == 0 ? (P - 1) / R + (q == 0 ? 1 : 0) : P / R));
    Hdr h;
    hdr_init(h, CLS_DM, "da");
    memcpy(d->hdr_, &h, sizeof(h));
    d->data = &d->da;
    d->da.e = nullptr;
    d->da.ne = 0;
    d->da.elementtype = DMDA_ELEMENT_P1;  // PETSc's default; the reference sets Q1 itself
    d->M = M;
    d->N = N;
    d->P = P;
    d->dof = dof;
    d->sw = s;
    d->have_box = false;
    const double b[6] = {0, 1, 0, 1, 0, 1};
    memcpy(d->box, b, sizeof(b));
    d->coords = nullptr;
    d->own[0] = M;
    d->own[1] = N;
    d->own[2] = P;
    d->uses_grid = false;
    if (mesh.nx == 0) {  // the first DMDA of a program is the node mesh (TopOpt.cc:225-262)
        mesh.nx = M;
        mesh.ny = N;
        mesh.nz = P;
    }
    mesh.users++;
    *da = d;
    return 0;
}
PetscErrorCode DMSetFromOptions(DM) { return 0; }
PetscErrorCode DMSetUp(DM) { return 0; }
PetscErrorCode DMDASetUniformCoordinates(DM da, PetscReal x0, PetscReal x1, PetscReal y0, PetscReal y1, PetscReal z0,
                                         PetscReal z1) {
    DMFull *d = F(da);
    const double b[6] = {x0, x1, y0, y1, z0, z1};
    memcpy(d->box, b, sizeof(b));
    d->have_box = true;
    if (d->coords) {
        VecDestroy(&d->coords);
    }
    if (is_nodal(d) && !mesh.have_box && !mesh.g) {  // element size of the mesh: first nodal DM with coordinates
        memcpy(mesh.box, b, sizeof(b));
        mesh.have_box = true;
    }
    return 0;
}
PetscErrorCode DMDASetElementType(DM da, DMDAElementType t) {
    F(da)->da.elementtype = t;
    return 0;
}
PetscErrorCode DMDAGetInfo(DM da, PetscInt *dim, PetscInt *M, PetscInt *N, PetscInt *P, PetscInt *m, PetscInt *n,
                           PetscInt *p, PetscInt *dof, PetscInt *s, DMBoundaryType *bx, DMBoundaryType *by,
                           DMBoundaryType *bz, DMDAStencilType *st) {
    DMFull *d = F(da);
    if (dim) *dim = 3;
    if (M) *M = d->M;
    if (N) *N = d->N;
    if (P) *P = d->P;
    if (m) *m = 1;
    if (n) *n = 1;
    if (p) *p = job_size();
    if (dof) *dof = d->dof;
    if (s) *s = d->sw;
    if (bx) *bx = DM_BOUNDARY_NONE;
    if (by) *by = DM_BOUNDARY_NONE;
    if (bz) *bz = DM_BOUNDARY_NONE;
    if (st) *st = DMDA_STENCIL_BOX;
    return 0;
}
PetscErrorCode DMDAGetCorners(DM da, PetscInt *x, PetscInt *y, PetscInt *z, PetscInt *m, PetscInt *n, PetscInt *p) {
    DMFull *d = F(da);
    const ZBox b = zbox(d, d->sw);
    if (x) *x = 0;
    if (y) *y = 0;
    if (z) *z = b.zs;
    if (m) *m = d->M;
    if (n) *n = d->N;
    if (p) *p = b.zm;
    return 0;
}
PetscErrorCode DMDAGetGhostCorners(DM da, PetscInt *x, PetscInt *y, PetscInt *z, PetscInt *m, PetscInt *n, PetscInt *p) {
    DMFull *d = F(da);  // non-periodic, x and y unsplit: ghost points only towards the neighbouring slabs
    const ZBox b = zbox(d, d->sw);
    if (x) *x = 0;
    if (y) *y = 0;
    if (z) *z = b.gzs;
    if (m) *m = d->M;
    if (n) *n = d->N;
    if (p) *p = b.gzm;
    return 0;
}
PetscErrorCode DMDAGetOwnershipRanges(DM da, const PetscInt *lx[], const PetscInt *ly[], const PetscInt *lz[]) {
    DMFull *d = F(da);
    if (lx) *lx = &d->own[0];
    if (ly) *ly = &d->own[1];
    if (lz) *lz = d->lzv.data();
    return 0;
}
PetscErrorCode DMDAGetLocalInfo(DM da, DMDALocalInfo *i) {
    DMFull *d = F(da);
    memset(i, 0, sizeof(*i));
    i->dim = 3;
    i->dof = d->dof;
    i->sw = d->sw;
    const ZBox b = zbox(d, d->sw);
    i->mx = i->xm = i->gxm = d->M;
    i->my = i->ym = i->gym = d->N;
    i->mz = d->P;
    i->zs = b.zs;
    i->zm = b.zm;
    i->gzs = b.gzs;
    i->gzm = b.gzm;
    i->st = DMDA_STENCIL_BOX;
    i->da = da;
    return 0;
}
PetscErrorCode DMDAGetElements(DM da, PetscInt *nel, PetscInt *nen, const PetscInt *e[]) {
    DMFull *d = F(da);
    if (!d->da.e) {  // hexahedra, DMDA natural order (the numbering of LinearElasticity.cc:819-826), ghosted local node numbers
        const ZBox b = zbox(d, d->sw);
        const PetscInt ex = d->M - 1, ey = d->N - 1;
        const PetscInt k0 = (b.zs != b.gzs ? b.zs - 1 : b.zs) - b.gzs, ez = k0 + (b.zs + b.zm - 1 - (b.zs != b.gzs ? b.zs - 1 : b.zs));
        d->da.ne = ex * ey * (ez - k0);
        d->da.e = (PetscInt *)malloc(sizeof(PetscInt) * (size_t)(1 + 8 * (long)d->da.ne));
        long c = 0;
        for (PetscInt k = k0; k < ez; k++)
            for (PetscInt j = 0; j < ey; j++)
                for (PetscInt i = 0; i < ex; i++) {
                    const PetscInt n0 = i + d->M * (j + d->N * k), dz = d->M * d->N;
                    const PetscInt cell[8] = {n0, n0 + 1, n0 + 1 + d->M, n0 + d->M, n0 + dz, n0 + 1 + dz, n0 + 1 + d->M + dz, n0 + d->M + dz};
                    for (int q = 0; q < 8; q++) d->da.e[c++] = cell[q];
                }
    }
    *nel = d->da.ne;
    *nen = 8;
    *e = d->da.e;
    return 0;
}
PetscErrorCode DMDARestoreElements(DM, PetscInt *, PetscInt *, const PetscInt *[]) { return 0; }
PetscErrorCode DMGetCoordinatesLocal(DM da, Vec *c) {
    DMFull *d = F(da);
    if (!d->coords) {  // the ghosted local box of this rank
        const ZBox b = zbox(d, d->sw);
        const long n = (long)d->M * d->N * b.gzm;
        int rc = vec_create(3 * n, true, da, &d->coords);
        if (rc) return rc;
        const double hx = d->M > 1 ? (d->box[1] - d->box[0]) / (d->M - 1) : 0.0, hy = d->N > 1 ? (d->box[3] - d->box[2]) / (d->N - 1) : 0.0,
                     hz = d->P > 1 ? (d->box[5] - d->box[4]) / (d->P - 1) : 0.0;
        double *p = d->coords->host.data();
        for (PetscInt k = b.gzs; k < b.gzs + b.gzm; k++)
            for (PetscInt j = 0; j < d->N; j++)
                for (PetscInt i = 0; i < d->M; i++) {  // DMDASetUniformCoordinates: xmin + i * h
                    *p++ = d->box[0] + hx * i;
                    *p++ = d->box[2] + hy * j;
                    *p++ = d->box[4] + hz * k;
                }
    }
    *c = d->coords;
    return 0;
}
PetscErrorCode DMGetLocalToGlobalMapping(DM, ISLocalToGlobalMapping *m) {
    static _p_ISLocalToGlobalMapping identity;
    *m = &identity;
    return 0;
}
static int dm_vector(DM da, bool local, Vec *v) {
    DMFull *d = F(da);
    if (!is_nodal(d) && !is_elem(d)) return sup("vector on a DMDA that is neither the node mesh nor its element mesh");
    d->uses_grid = true;
    const long per = (long)d->dof * d->M * d->N;  // entries per z-plane
    const long nglob = per * d->P;
    if (job_size() == 1) return vec_create_layout(nglob, 0, nglob, nglob, 0, local, false, da, v);
    const ZBox own = zbox(d, 0);
    if (is_elem(d)) {
        if (local) return sup("DMCreateLocalVector on the element mesh across ranks");
        return vec_create_layout(per * own.zm, 0, per * own.zm, nglob, per * own.zs, false, false, da, v);
    }
    const ZBox gb = zbox(d, 1);  // the library's slab layout: one ghost plane towards each neighbour
    if (d->sw != 1) return sup("node mesh with a stencil width other than 1 across ranks");
    if (local) return vec_create_layout(per * gb.gzm, 0, per * gb.gzm, nglob, per * gb.gzs, true, false, da, v);
    return vec_create_layout(per * gb.gzm, per * (own.zs - gb.gzs), per * own.zm, nglob, per * own.zs, false, false, da, v);
}
PetscErrorCode DMCreateGlobalVector(DM da, Vec *v) { return dm_vector(da, false, v); }
PetscErrorCode DMCreateLocalVector(DM da, Vec *v) { return dm_vector(da, true, v); }
// global -> ghosted local: both are slab arrays; copy the slab, then fetch the ghost planes from the neighbours
PetscErrorCode DMGlobalToLocalBegin(DM da, Vec g, InsertMode, Vec l) {
    if (g == l) return 0;
    if (g->n_alloc != l->n_alloc) return PETSC_ERR_ARG_WRONG;
    const double *pg = bin(g);
    int rc = tp_vec_axpby(mesh.g, bout(l), 1.0, pg, 0.0, l->n_alloc);
    if (!rc && job_size() > 1) rc = tp_grid_halo_nodes(mesh.g, l->d, (int)F(da)->dof);
    return rc;
}
PetscErrorCode DMGlobalToLocalEnd(DM, Vec, InsertMode, Vec) { return 0; }
PetscErrorCode DMCoarsenHierarchy(DM da, PetscInt nlevels, DM dac[]) {
    DMFull *f = F(da);
    PetscInt M = f->M, N = f->N, P = f->P;
    for (PetscInt l = 0; l < nlevels; l++) {
        if ((M - 1) % 2 || (N - 1) % 2 || (P - 1) % 2) return sup("DMCoarsenHierarchy: element counts not divisible by 2 (TopOpt.cc:183-201)");
        M = (M - 1) / 2 + 1;
        N = (N - 1) / 2 + 1;
        P = (P - 1) / 2 + 1;
        int rc = DMDACreate3d(0, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DM_BOUNDARY_NONE, DMDA_STENCIL_BOX, M, N, P, 1, 1, job_size(), f->dof, f->sw, 0, 0, 0, &dac[l]);
        if (rc) return rc;
    }
    return 0;
}
PetscErrorCode DMCreateInterpolation(DM dac, DM daf, Mat *P, Vec *scale) {
    DMFull *c = F(dac), *f = F(daf);
    if ((f->M - 1) != 2 * (c->M - 1) || (f->N - 1) != 2 * (c->N - 1) || (f->P - 1) != 2 * (c->P - 1))
        return sup("DMCreateInterpolation: only factor-2 trilinear (Q1) interpolation between DMDAs");
    *P = mat_new(K_INTERP, daf, (long)f->dof * f->M * f->N * zbox(f, 0).zm, (long)c->dof * c->M * c->N * zbox(c, 0).zm, "q1interp");
    if (scale) *scale = nullptr;
    return 0;
}
PetscErrorCode DMCreateMatrix(DM da, Mat *A) {
    DMFull *d = F(da);
    const long n = (long)d->dof * d->M * d->N * zbox(d, 0).zm;  // local rows
    if (is_nodal(d) && d->dof == 3) {
        *A = mat_new(K_ELAST, da, n, n, "topopt-elasticity");
    } else if (is_nodal(d) && d->dof == 1) {
        *A = mat_new(K_HELM, da, n, n, "topopt-helmholtz");
        g_last_helm = *A;
    } else if (is_elem(d) && d->dof == 1) {
        *A = mat_new(K_CONE, da, n, n, "topopt-conefilter");
    } else {
        return sup("DMCreateMatrix: dof-3 / dof-1 node mesh or dof-1 element mesh only");
    }
    return 0;
}
PetscErrorCode DMDestroy(DM *da) {
    if (da && *da) {
        DMFull *d = F(*da);
        if (d->coords) VecDestroy(&d->coords);
        free(d->da.e);
        delete d;
        *da = nullptr;
        if (--mesh.users == 0 && mesh.g) {
            tp_grid_destroy(mesh.g);
            mesh = Mesh();
        }
    }
    return 0;
}

}  // extern "C"

// shim/ext.cc -- PETSc-named surface of include/petsc_compat/petsc.h, part "ext" (see shim/internal.h)
#include "internal.h"

extern "C" {

// =============================================================================================== extension calls
PetscErrorCode MatCreateTopOptElasticity(DM da, PetscScalar nu, PetscInt nlvls, Mat *K) {
    DMFull *d = F(da);
    if (!da || d->dof != 3 || !is_nodal(d)) return PETSC_ERR_ARG_WRONG;
    if (d->have_box && !mesh.g) {
        memcpy(mesh.box, d->box, sizeof(mesh.box));
        mesh.have_box = true;
    }
    int rc = ensure_grid();
    if (rc) return rc;
    tp_solver_opts o;
    tp_solver_default_opts(&o);
    o.nlvls = nlvls;
    o.nu = nu;
    const long n = 3L * d->M * d->N * zbox(d, 0).zm;
    Mat A = mat_new(K_EXT_ELAST, da, n, n, "topopt-elasticity");
    rc = tp_elasticity_create(&A->e, mesh.g, &o);
    if (rc) {
        delete A;
        return rc;
    }
    *K = A;
    return 0;
}
PetscErrorCode MatTopOptCantilever(Mat K, Vec N, Vec RHS) {
    if (!K || K->kind != K_EXT_ELAST || N->n != K->n_rows || RHS->n != K->n_rows) return PETSC_ERR_ARG_WRONG;
    K->have_bc = true;
    return tp_elasticity_cantilever(K->e, bout(N), bout(RHS));  // also registers N
}
PetscErrorCode MatTopOptSetDirichlet(Mat K, Vec N) {
    if (!K || K->kind != K_EXT_ELAST || N->n != K->n_rows) return PETSC_ERR_ARG_WRONG;
    K->have_bc = true;
    int rc = job_size() > 1 ? tp_grid_halo_nodes(mesh.g, binout(N), 3) : 0;
    return rc ? rc : tp_elasticity_set_bc(K->e, bin(N));
}
PetscErrorCode MatTopOptAssemble(Mat K, Vec xPhys, PetscScalar Emin, PetscScalar Emax, PetscScalar penal) {
    if (!K || K->kind != K_EXT_ELAST) return PETSC_ERR_ARG_WRONG;
    if (!K->have_bc) return PETSC_ERR_ORDER;
    K->ext_assembled = true;
    return tp_elasticity_assemble(K->e, din(xPhys), Emin, Emax, penal);
}
PetscErrorCode MatTopOptComplianceSensitivity(Mat K, Vec U, Vec xPhys, PetscScalar Emin, PetscScalar Emax,
                                              PetscScalar penal, PetscScalar volfrac, PetscScalar *fx, PetscScalar *gx,
                                              Vec dfdx, Vec dgdx) {
    if (!K || (K->kind != K_EXT_ELAST && K->kind != K_ELAST) || !K->e) return PETSC_ERR_ARG_WRONG;
    const double *pu = binout(U), *px = din(xPhys);
    return tp_elasticity_objective(K->e, pu, px, Emin, Emax, penal, volfrac, fx, gx, dfdx ? dout(dfdx) : nullptr,
                                   dgdx ? dout(dgdx) : nullptr);
}
PetscErrorCode MatCreateTopOptFilter(DM da, PetscInt filterType, PetscScalar R, Mat *H, Vec *Hs) {
    DMFull *d = F(da);
    if (!da || filterType < 0 || filterType > 2 || !is_nodal(d)) return sup("MatCreateTopOptFilter: types 0, 1, 2 on the node mesh");
    if (d->have_box && !mesh.g) {
        memcpy(mesh.box, d->box, sizeof(mesh.box));
        mesh.have_box = true;
    }
    int rc = ensure_grid();
    if (rc) return rc;
    const long nel = tp_grid_local_elems(mesh.g);
    Mat A = mat_new(K_EXT_FILTER, da, nel, nel, "topopt-filter");
    A->coneR = filterType == 2 ? -1.0 : R;
    rc = tp_filter_create(&A->f, mesh.g, filterType, R, nullptr);
    if (rc) {
        delete A;
        return rc;
    }
    if (Hs) {
        const long per = (long)(d->M - 1) * (d->N - 1);
        rc = vec_create_layout(nel, 0, nel, per * (d->P - 1), per * ((d->P - 1) / job_size()) * job_rank(), job_size() == 1, false, nullptr, Hs);
        if (!rc) rc = filterType == 2 ? VecSet(*Hs, 1.0) : tp_filter_get_hs(A->f, dout(*Hs));
    }
    *H = A;
    return rc;
}

}  // extern "C"

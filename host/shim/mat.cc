// shim/mat.cc -- PETSc-named surface of include/petsc_compat/petsc.h, part "mat" (see shim/internal.h)
#include "internal.h"

extern "C" {

// =============================================================================================== Mat
PetscErrorCode MatCreateAIJ(MPI_Comm, PetscInt m, PetscInt n, PetscInt, PetscInt, PetscInt, const PetscInt[], PetscInt,
                            const PetscInt[], Mat *A) {
    // the only AIJ matrix of the path: T (nodes x elements, PDEFilter.cc:143-170)
    // local sizes: this rank's owned nodes x owned elements
    const int R = job_size(), r = job_rank();
    const long ezl = (mesh.nz - 1) / R, own_planes = R == 1 ? mesh.nz : ezl + (r == 0 ? 1 : 0);
    if (m != (PetscInt)((long)mesh.nx * mesh.ny * own_planes) || n != (PetscInt)((long)(mesh.nx - 1) * (mesh.ny - 1) * ezl))
        return sup("MatCreateAIJ: only the nodes x elements transfer matrix of the PDE filter");
    *A = mat_new(K_TMAT, nullptr, m, n, "topopt-elem2node");
    return 0;
}
PetscErrorCode MatSetLocalToGlobalMapping(Mat, ISLocalToGlobalMapping, ISLocalToGlobalMapping) { return 0; }
PetscErrorCode MatZeroEntries(Mat A) {
    A->ncalls = 0;
    if (A->kind == K_ELAST) std::fill(A->E.begin(), A->E.end(), 0.0);
    return 0;
}
PetscErrorCode MatSetValuesLocal(Mat A, PetscInt nrow, const PetscInt irow[], PetscInt ncol, const PetscInt icol[],
                                 const PetscScalar y[], InsertMode addv) {
    switch (A->kind) {
    case K_ELAST: {  // AssembleStiffnessMatrix, LinearElasticity.cc:510-524: ke = KE * dens, ADD_VALUES
        if (nrow != 24 || ncol != 24 || addv != ADD_VALUES) return sup("dof-3 matrix: 24x24 ADD_VALUES element blocks only");
        DMFull *d = F(A->dm);
        // local numbering: node planes counted from the first stored (ghost) plane, whose element layer is this rank's first
        const PetscInt ex = d->M - 1, ey = d->N - 1, ez = (d->P - 1) / job_size();
        const PetscInt n0 = irow[0] / 3, i = n0 % d->M, j = (n0 / d->M) % d->N, k = n0 / (d->M * d->N);
        if (irow[0] % 3 || i >= ex || j >= ey || k >= ez || icol[0] != irow[0] || irow[3] != 3 * (n0 + 1))
            return sup("dof-3 matrix: rows are not the 24 dofs of a hexahedron in DMDA order");
        const long el = (long)i + (long)ex * (j + (long)ey * k);
        if (A->E.empty()) A->E.assign((size_t)ex * ey * ez, 0.0);
        if (A->ref.empty()) A->ref.assign(y, y + 576);
        const double s = y[0] / A->ref[0];
        for (int q : {0, 1, 25, 300, 575})
            if (fabs(y[q] - s * A->ref[q]) > 1e-12 * fabs(s) * (fabs(A->ref[0]) + fabs(A->ref[q])))
                return sup("dof-3 matrix: element blocks are not multiples of one element matrix");
        if (getenv("TP_SHIM_VERIFY")) {  // the whole block and all 24 indices, not samples
            for (int q = 0; q < 576; q++)
                if (fabs(y[q] - s * A->ref[q]) > 1e-12 * fabs(s) * (fabs(A->ref[0]) + fabs(A->ref[q])))
                    return sup("dof-3 matrix (verify): an element block is not a multiple of the first one");
            const PetscInt dz = d->M * d->N;
            const PetscInt cell[8] = {n0, n0 + 1, n0 + 1 + d->M, n0 + d->M, n0 + dz, n0 + 1 + dz, n0 + 1 + d->M + dz, n0 + d->M + dz};
            for (int a = 0; a < 8; a++)
                for (int c = 0; c < 3; c++)
                    if (irow[3 * a + c] != 3 * cell[a] + c || icol[3 * a + c] != irow[3 * a + c])
                        return sup("dof-3 matrix (verify): rows/columns are not the hexahedron's dofs in the reference's corner order");
            A->nverified++;
        }
        A->E[(size_t)el] += s;
        A->ncalls++;
        A->assembled_since_setup = true;
        return 0;
    }
    case K_HELM:  // PDEFilt::MatAssemble, PDEFilter.cc:257-260: the same KF for every element
        if (nrow != 8 || ncol != 8 || addv != ADD_VALUES) return sup("dof-1 node matrix: 8x8 ADD_VALUES element blocks only");
        if (A->ref.empty()) A->ref.assign(y, y + 64);
        else if (memcmp(A->ref.data(), y, sizeof(double) * 64) != 0) return sup("dof-1 node matrix: variable coefficients");
        A->ncalls++;
        return 0;
    case K_TMAT:  // :262: T(8 nodes, element) = TF = 1/8
        if (nrow != 8 || ncol != 1 || addv != ADD_VALUES) return sup("transfer matrix: 8x1 ADD_VALUES blocks only");
        for (int q = 0; q < 8; q++)
            if (y[q] != 0.125) return sup("transfer matrix: entries other than 1/8");
        A->ncalls++;
        return 0;
    case K_CONE:  // Filter::SetUp, Filter.cc:417-433: H(row, col) = R - dist, INSERT_VALUES; H(row, row) = R
        if (nrow != 1 || ncol != 1 || addv != INSERT_VALUES) return sup("element matrix: 1x1 INSERT_VALUES entries only");
        if (irow[0] == icol[0]) {
            if (A->coneR == 0.0) A->coneR = y[0];
            else if (A->coneR != y[0]) return sup("element matrix: varying diagonal (not a cone filter of one radius)");
        } else if (A->coneR != 0.0 && !(y[0] > 0.0 && y[0] < A->coneR)) {
            return sup("element matrix: off-diagonal weight outside (0, R)");
        }
        if (getenv("TP_SHIM_VERIFY") && job_size() == 1) {  // keep the caller's matrix to check the device filter against it
            A->hrow.push_back((int)irow[0]);
            A->hcol.push_back((int)icol[0]);
            A->hval.push_back(y[0]);
        }
        A->ncalls++;
        return 0;
    default:
        return sup("MatSetValuesLocal on this matrix");
    }
}
PetscErrorCode MatAssemblyBegin(Mat, MatAssemblyType) { return 0; }
PetscErrorCode MatAssemblyEnd(Mat A, MatAssemblyType) {
    if (A->kind == K_CONE && !A->f) {
        if (A->coneR <= 0.0) return sup("element matrix without diagonal entries");
        int rc = ensure_grid();
        if (!rc) rc = tp_filter_create(&A->f, mesh.g, 1, A->coneR, nullptr);
        if (!rc && !A->hval.empty()) {
            // TP_SHIM_VERIFY=1: only the radius was taken from the caller's entries -- check that the device filter IS the
            // matrix the caller assembled (Filter.cc:417-433: the reference's own distances and weights): H x for a
            // pseudo-random x, entry by entry on the host, against tp_filter_mult_h
            const long n = A->n_rows;
            std::vector<double> x((size_t)n), yh((size_t)n, 0.0), yd((size_t)n);
            uint64_t st = 0x9E3779B97F4A7C15ULL;
            for (long i = 0; i < n; i++) {
                st = st * 6364136223846793005ULL + 1442695040888963407ULL;
                x[(size_t)i] = (double)(st >> 11) / 9007199254740992.0;
            }
            for (size_t e = 0; e < A->hval.size(); e++) yh[(size_t)A->hrow[e]] += A->hval[e] * x[(size_t)A->hcol[e]];
            double *dx = nullptr, *dy = nullptr;
            rc = tp_malloc((void **)&dx, sizeof(double) * (size_t)n) || tp_malloc((void **)&dy, sizeof(double) * (size_t)n);
            if (!rc) rc = tp_memcpy_h2d(dx, x.data(), sizeof(double) * (size_t)n);
            if (!rc) rc = tp_filter_mult_h(A->f, dx, dy);
            if (!rc) rc = tp_sync(mesh.g);
            if (!rc) rc = tp_memcpy_d2h(yd.data(), dy, sizeof(double) * (size_t)n);
            if (dx) tp_free(dx);
            if (dy) tp_free(dy);
            if (rc) return rc;
            double dev = 0.0, scale = 0.0;
            for (long i = 0; i < n; i++) {
                dev = fmax(dev, fabs(yh[(size_t)i] - yd[(size_t)i]));
                scale = fmax(scale, fabs(yh[(size_t)i]));
            }
            printf("[petsc-compat] verified the cone filter against the %zu inserted entries: max |H x - device| / max |H x| = %.3e\n",
                   A->hval.size(), dev / scale);
            if (!(dev <= 1e-12 * scale)) return sup("element matrix: the inserted entries are not the cone filter of their diagonal's radius on this mesh");
            A->hrow.clear();
            A->hcol.clear();
            A->hval.clear();
            A->hrow.shrink_to_fit();
            A->hcol.shrink_to_fit();
            A->hval.shrink_to_fit();
        }
        return rc;
    }
    return 0;
}
PetscErrorCode MatDiagonalScale(Mat A, Vec l, Vec r) {  // K = N K N, LinearElasticity.cc:533
    if (A->kind != K_ELAST || l != r || !l || l->n != A->n_rows) return sup("MatDiagonalScale: (K, N, N) on the stiffness matrix only");
    if (!A->Nvec) {
        int rc = VecDuplicate(l, &A->Nvec);
        if (rc) return rc;
    }
    A->have_bc = true;
    A->assembled_since_setup = true;
    return VecCopy(l, A->Nvec);
}
PetscErrorCode MatDiagonalSet(Mat A, Vec D, InsertMode mode) {  // K += I - N, :534-538
    if (A->kind != K_ELAST || mode != ADD_VALUES || !A->have_bc) return sup("MatDiagonalSet: (K, I - N, ADD_VALUES) after MatDiagonalScale only");
    double sd = 0.0, sn = 0.0;
    int rc = VecSum(D, &sd);
    if (!rc) rc = VecSum(A->Nvec, &sn);
    if (rc) return rc;
    if (sd + sn != (double)A->Nvec->nglob) return sup("MatDiagonalSet: the vector is not I - N");
    return 0;
}
PetscErrorCode MatMult(Mat A, Vec x, Vec y) {
    if (!A || !x || !y || x->n != A->n_cols || y->n != A->n_rows) return PETSC_ERR_ARG_WRONG;
    switch (A->kind) {
    case K_ELAST: {
        int rc = ensure_elasticity(A);
        if (rc) return rc;
        const double *px = binout(x);  // (the library refreshes the ghost planes of its input)
        return tp_elasticity_apply(A->e, px, bout(y));
    }
    case K_EXT_ELAST:
        {
        if (!A->ext_assembled) return PETSC_ERR_ORDER;
        const double *px = binout(x);
        return tp_elasticity_apply(A->e, px, bout(y));
    }
    case K_CONE:
        {
        if (!A->f) return PETSC_ERR_ORDER;
        const double *px = din(x);
        return tp_filter_mult_h(A->f, px, dout(y));
    }
    case K_EXT_FILTER:
        if (A->coneR < 0.0) {  // PDE filter as one operator
            const double *px = din(x);
            double *py = dout(y);
            return tp_filter_project(A->f, px, py, py, 0, 0.0, 0.0) ? PETSC_ERR_ARG_WRONG : 0;
        }
        {
            const double *px = din(x);
            return tp_filter_mult_h(A->f, px, dout(y));
        }
    case K_HELM: {
        int rc = ensure_pdefilter(A);
        if (rc) return rc;
        const double *px = binout(x);
        return tp_pdefilter_apply(A->f, px, bout(y));
    }
    case K_TMAT: {
        if (!g_last_helm) return PETSC_ERR_ORDER;
        int rc = ensure_pdefilter(g_last_helm);
        if (rc) return rc;
        const double *px = din(x);
        return tp_pdefilter_elem_to_node(g_last_helm->f, px, bout(y));
    }
    default:
        return sup("MatMult on this matrix");
    }
}
PetscErrorCode MatMultTranspose(Mat A, Vec x, Vec y) {
    if (!A || A->kind != K_TMAT) {
        if (A && (A->kind == K_ELAST || A->kind == K_HELM || A->kind == K_CONE || A->kind == K_EXT_ELAST)) return MatMult(A, x, y);  // symmetric
        return sup("MatMultTranspose on this matrix");
    }
    if (x->n != A->n_rows || y->n != A->n_cols || !g_last_helm) return PETSC_ERR_ARG_WRONG;
    int rc = ensure_pdefilter(g_last_helm);
    if (rc) return rc;
    const double *px = binout(x);  // its ghost planes are refreshed
    return tp_pdefilter_node_to_elem(g_last_helm->f, px, dout(y));
}
PetscErrorCode MatDestroy(Mat *A) {
    if (A && *A) {
        if (--(*A)->h.refct <= 0) {
            if ((*A)->e) tp_elasticity_destroy((*A)->e);
            if ((*A)->f) tp_filter_destroy((*A)->f);
            if ((*A)->dE) tp_free((*A)->dE);
            if ((*A)->Nvec) VecDestroy(&(*A)->Nvec);
            if (g_last_helm == *A) g_last_helm = nullptr;
            delete *A;
        }
        *A = nullptr;
    }
    return 0;
}

}  // extern "C"

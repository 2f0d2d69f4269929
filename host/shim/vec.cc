// shim/vec.cc -- PETSc-named surface of include/petsc_compat/petsc.h, part "vec" (see shim/internal.h)
#include "internal.h"

extern "C" {

// =============================================================================================== Vec
PetscErrorCode VecDuplicate(Vec v, Vec *nv) {
    return vec_create_layout(v->n_alloc, v->off, v->n, v->nglob, v->goff, v->is_local, v->d == nullptr, v->dm, nv);
}
PetscErrorCode VecDuplicateVecs(Vec v, PetscInt m, Vec *V[]) {
    *V = (Vec *)malloc(sizeof(Vec) * (size_t)(m > 0 ? m : 1));
    for (PetscInt i = 0; i < m; i++) {
        int rc = VecDuplicate(v, &(*V)[i]);
        if (rc) return rc;
    }
    return 0;
}
PetscErrorCode VecDestroyVecs(PetscInt m, Vec *V[]) {
    if (V && *V) {
        for (PetscInt i = 0; i < m; i++) VecDestroy(&(*V)[i]);
        free(*V);
        *V = nullptr;
    }
    return 0;
}
PetscErrorCode VecDestroy(Vec *v) {
    if (v && *v) {
        if (--(*v)->h.refct <= 0) {
            if ((*v)->d) tp_free((*v)->d);
            delete *v;
        }
        *v = nullptr;
    }
    return 0;
}
PetscErrorCode VecSet(Vec v, PetscScalar a) {
    if (!v->d) {
        std::fill(v->host.begin(), v->host.end(), a);
        return 0;
    }
    return tp_vec_set(mesh.g, dout(v), a, v->n);
}
PetscErrorCode VecCopy(Vec x, Vec y) {
    if (x->n != y->n) return PETSC_ERR_ARG_WRONG;
    if (x == y) return 0;
    const double *px = din(x);
    return tp_vec_axpby(mesh.g, dout(y), 1.0, px, 0.0, y->n);
}
PetscErrorCode VecScale(Vec v, PetscScalar a) { return tp_vec_scale(mesh.g, dinout(v), a, v->n); }
PetscErrorCode VecAXPY(Vec y, PetscScalar a, Vec x) {
    if (x->n != y->n) return PETSC_ERR_ARG_WRONG;
    const double *px = din(x);
    return tp_vec_axpby(mesh.g, dinout(y), a, px, 1.0, y->n);
}
PetscErrorCode VecAXPBY(Vec y, PetscScalar a, PetscScalar b, Vec x) {
    if (x->n != y->n) return PETSC_ERR_ARG_WRONG;
    const double *px = din(x);
    return tp_vec_axpby(mesh.g, dinout(y), a, px, b, y->n);
}
PetscErrorCode VecAXPBYPCZ(Vec z, PetscScalar alpha, PetscScalar beta, PetscScalar gamma, Vec x, Vec y) {  // z = a x + b y + c z
    const double *px = din(x), *py = din(y);
    double *pz = gamma == 0.0 ? dout(z) : dinout(z);
    int rc = tp_vec_axpby(mesh.g, pz, alpha, px, gamma, z->n);
    return rc ? rc : tp_vec_axpby(mesh.g, pz, beta, py, 1.0, z->n);
}
PetscErrorCode VecPointwiseMult(Vec w, Vec x, Vec y) {
    if (w->n != x->n || w->n != y->n) return PETSC_ERR_ARG_WRONG;
    const double *px = din(x), *py = din(y);
    return tp_vec_pointwise(mesh.g, (w == x || w == y) ? dinout(w) : dout(w), px, py, 0, w->n);
}
PetscErrorCode VecPointwiseDivide(Vec w, Vec x, Vec y) {
    if (w->n != x->n || w->n != y->n) return PETSC_ERR_ARG_WRONG;
    const double *px = din(x), *py = din(y);
    return tp_vec_pointwise(mesh.g, (w == x || w == y) ? dinout(w) : dout(w), px, py, 1, w->n);
}
PetscErrorCode VecDot(Vec x, Vec y, PetscScalar *val) {
    if (x->n != y->n) return PETSC_ERR_ARG_WRONG;
    return tp_vec_dot(mesh.g, din(x), din(y), x->n, val);
}
PetscErrorCode VecNorm(Vec x, NormType type, PetscReal *val) {
    if (type != NORM_2) return sup("VecNorm: NORM_2 only");
    double s = 0.0;
    const double *px = din(x);
    int rc = tp_vec_dot(mesh.g, px, px, x->n, &s);
    *val = std::sqrt(s);
    return rc;
}
PetscErrorCode VecSum(Vec x, PetscScalar *sum) { return tp_vec_dot(mesh.g, din(x), nullptr, x->n, sum); }
static int vec_extreme(Vec x, bool want_max, PetscInt *p, PetscReal *val) {
    int rc = vec_pull(x);
    const double *h = x->host.data() + x->off;
    long at = 0;
    for (long i = 1; i < x->n; i++)
        if (want_max ? h[i] > h[at] : h[i] < h[at]) at = i;
    double v = x->n ? h[at] : (want_max ? -1e300 : 1e300);
    if (!x->is_local && job_size() > 1) {
        if (p) return sup("VecMax / VecMin: the location of the extremum across ranks");
        if (comm_ready()) return PETSC_ERR_LIB;
        slab_detail::host_reduce(&sc, &v, 1, want_max ? 1 : 2);
    }
    if (p) *p = (PetscInt)at;
    if (val) *val = v;
    return rc;
}
PetscErrorCode VecMax(Vec x, PetscInt *p, PetscReal *val) { return vec_extreme(x, true, p, val); }
PetscErrorCode VecMin(Vec x, PetscInt *p, PetscReal *val) { return vec_extreme(x, false, p, val); }
PetscErrorCode VecGetSize(Vec x, PetscInt *n) {
    *n = (PetscInt)(x->is_local ? x->n : x->nglob);
    return 0;
}
PetscErrorCode VecGetLocalSize(Vec x, PetscInt *n) {
    *n = (PetscInt)x->n;
    return 0;
}
PetscErrorCode VecGetArray(Vec x, PetscScalar **a) {
    int rc = vec_pull(x);
    if (x->d) x->dev_valid = false;  // the caller may write through the pointer, now or later
    *a = x->host.data() + x->off;
    return rc;
}
PetscErrorCode VecRestoreArray(Vec, PetscScalar **a) {  // nothing to copy: the next device use pushes the mirror
    if (a) *a = nullptr;
    return 0;
}
PetscErrorCode VecGetArrays(const Vec x[], PetscInt n, PetscScalar **a[]) {
    PetscScalar **q = (PetscScalar **)malloc(sizeof(PetscScalar *) * (size_t)(n > 0 ? n : 1));
    for (PetscInt i = 0; i < n; i++) {
        int rc = VecGetArray(x[i], &q[i]);
        if (rc) return rc;
    }
    *a = q;
    return 0;
}
PetscErrorCode VecRestoreArrays(const Vec[], PetscInt, PetscScalar **a[]) {
    if (a && *a) {
        free(*a);
        *a = nullptr;
    }
    return 0;
}
// local numbering = PETSc's ghosted local numbering of the DMDA = the position in the stored slab.  An entry set on a
// ghost node stays local: the owner sets the same entry itself (the reference's set-up loops run over all local nodes
// on every rank, LinearElasticity.cc:148-172), which is what VecAssembly would deliver.
PetscErrorCode VecSetValueLocal(Vec v, PetscInt row, PetscScalar value, InsertMode mode) {
    if (row < 0 || row >= v->n_alloc) return PETSC_ERR_ARG_OUTOFRANGE;
    int rc = vec_pull(v);
    if (rc) return rc;
    if (v->d) v->dev_valid = false;
    if (mode == ADD_VALUES) v->host[(size_t)row] += value;
    else v->host[(size_t)row] = value;
    return 0;
}
PetscErrorCode VecSetValue(Vec v, PetscInt row, PetscScalar value, InsertMode mode) {  // global index
    if (row < v->goff || row >= v->goff + v->n) return job_size() > 1 ? sup("VecSetValue on an entry of another rank") : PETSC_ERR_ARG_OUTOFRANGE;
    return VecSetValueLocal(v, (PetscInt)(row - v->goff + v->off), value, mode);
}
PetscErrorCode VecAssemblyBegin(Vec) { return 0; }
PetscErrorCode VecAssemblyEnd(Vec v) { return vec_push(v); }
PetscErrorCode VecSetRandom(Vec v, PetscRandom r) {
    int rc0 = vec_pull(v);
    if (rc0) return rc0;
    // drand48's linear congruential generator; one global sequence in natural order whatever the partition
    for (long i = 0; i < v->nglob; i++) {
        r->state = (r->state * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
        if (i >= v->goff && i < v->goff + v->n) v->host[(size_t)(i - v->goff + v->off)] = (double)r->state / (double)(1ULL << 48);
    }
    v->host_valid = true;
    v->dev_valid = false;
    return vec_push(v);
}
// PETSc binary Vec: big-endian int32 class id 1211214, int32 n, n big-endian doubles
PetscErrorCode VecView(Vec v, PetscViewer w) {
    int rc = vec_pull(v);
    if (rc) return rc;
    auto be32 = [&](uint32_t x) {
        unsigned char b[4] = {(unsigned char)(x >> 24), (unsigned char)(x >> 16), (unsigned char)(x >> 8), (unsigned char)x};
        fwrite(b, 1, 4, w->fp);
    };
    // every rank writes its own part at its place (natural ordering = the slab order); rank 0 also the header
    const long ng = v->is_local ? v->n : v->nglob, g0 = v->is_local ? 0 : v->goff;
    fseek(w->fp, (long)w->pos, SEEK_SET);
    if (job_rank() == 0 || v->is_local) {
        be32(1211214u);
        be32((uint32_t)ng);
    }
    fseek(w->fp, (long)(w->pos + 8 + 8 * g0), SEEK_SET);
    for (long i = 0; i < v->n; i++) {
        uint64_t u;
        memcpy(&u, &v->host[(size_t)(v->off + i)], 8);
        unsigned char b[8];
        for (int k = 0; k < 8; k++) b[k] = (unsigned char)(u >> (56 - 8 * k));
        fwrite(b, 1, 8, w->fp);
    }
    fflush(w->fp);
    w->pos += 8 + 8 * ng;
    return 0;
}
PetscErrorCode VecLoad(Vec v, PetscViewer w) {
    unsigned char b[8];
    auto be32 = [&](uint32_t *x) {
        if (fread(b, 1, 4, w->fp) != 4) return false;
        *x = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
        return true;
    };
    uint32_t cls, n;
    const long ng = v->is_local ? v->n : v->nglob, g0 = v->is_local ? 0 : v->goff;
    fseek(w->fp, (long)w->pos, SEEK_SET);
    if (!be32(&cls) || !be32(&n) || cls != 1211214u || (long)n != ng) return 79;  // PETSC_ERR_FILE_UNEXPECTED
    int rc0 = vec_pull(v);
    if (rc0) return rc0;
    fseek(w->fp, (long)(w->pos + 8 + 8 * g0), SEEK_SET);
    for (long i = 0; i < v->n; i++) {
        if (fread(b, 1, 8, w->fp) != 8) return 79;
        uint64_t u = 0;
        for (int k = 0; k < 8; k++) u = (u << 8) | b[k];
        memcpy(&v->host[(size_t)(v->off + i)], &u, 8);
    }
    w->pos += 8 + 8 * ng;
    v->host_valid = true;
    v->dev_valid = false;
    return vec_push(v);
}
PetscErrorCode VecTopOptGetDevicePointer(Vec x, PetscScalar **d) {
    *d = dinout(x);
    return 0;
}

}  // extern "C"

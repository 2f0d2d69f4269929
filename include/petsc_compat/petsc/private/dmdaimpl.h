/* petsc/private/dmdaimpl.h (compat) -- the two private structs the reference reaches into
 * (DMDAGetElements_3D: `DM_DA *da = (DM_DA *)dm->data; da->e; da->ne; da->elementtype`,
 * LinearElasticity.cc:785-839 and its three copies).  Only these members are part of the contract. */
#ifndef TOPOPT_PETSC_COMPAT_DMDAIMPL_H
#define TOPOPT_PETSC_COMPAT_DMDAIMPL_H
#include <petsc.h>

typedef struct {
    PetscInt *e;                 /* cached element connectivity (owned: freed by DMDestroy with PetscFree) */
    PetscInt ne;
    DMDAElementType elementtype;
} DM_DA;

struct _p_DM {
    void *hdr_[4];               /* object header of the compat layer */
    void *data;                  /* -> DM_DA */
};
#endif

/* petsc_shim.h -- kept for the thin adapter programs (host/shim_le.cc): the PETSc-named surface now lives in
 * include/petsc_compat/petsc.h (the header the reference's own sources compile against). */
#ifndef TOPOPT_PETSC_SHIM_H
#define TOPOPT_PETSC_SHIM_H
#include "petsc_compat/petsc.h"
#include "topopt_amd.h"
#endif

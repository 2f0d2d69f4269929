/* petsc_shim.h -- the one extension of libtopopt_petsc_shim.so beyond the PETSc names (used by host/ksp_probe.cc and
 * tests/test_cpp_host.py): what a configured KSP resolves to.  The PETSc-named surface itself lives in
 * include/petsc_compat/petsc.h (the header the reference's own sources compile against, host/shim/ behind it). */
#ifndef TOPOPT_PETSC_SHIM_H
#define TOPOPT_PETSC_SHIM_H
#include "petsc_compat/petsc.h"
#include "topopt_amd.h"
#ifdef __cplusplus
extern "C" {
#endif
/* what a configured KSP (types, tolerances, PCMG levels, options database) resolves to on the MI355X path -- the
 * tp_solver_opts the library is created with at KSPSetUp -- or PETSC_ERR_SUP with a message; no device is touched */
PetscErrorCode KSPCompatResolve(KSP ksp, tp_solver_opts *opts);
#ifdef __cplusplus
}
#endif
#endif

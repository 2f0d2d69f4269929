/* petsc_shim.h -- PETSc-named adapter over topopt_amd.h.
 *
 * The reference's hot path is written against PETSc 3.11 (Vec / Mat / KSP / DMDA).  This header declares, with
 * PETSc's names, argument order and error convention (PetscErrorCode, 0 = success, CHKERRQ-compatible), the part
 * of that surface that LinearElasticity.cc and Filter.cc touch per design iteration, implemented on the MI355X
 * library: vectors are HBM resident, `Mat` is the matrix-free elasticity operator or the cone filter, `KSP` is
 * PCG + geometric multigrid.  What PETSc builds by assembly has no counterpart here and is replaced
 * by three extension calls (MatCreateTopOpt*, MatTopOptAssemble, MatTopOptComplianceSensitivity), each citing the
 * reference lines it stands for.  One process; z-slab multi-GPU runs go through tp_comm / tp_grid_use_rccl.
 * Library: libtopopt_petsc_shim.so (host/petsc_shim.cc), links libtopopt_amd.so.
 */
#ifndef TOPOPT_PETSC_SHIM_H
#define TOPOPT_PETSC_SHIM_H
#include "topopt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef int PetscErrorCode;
typedef int PetscInt;
typedef double PetscScalar;
typedef double PetscReal;
typedef enum { PETSC_FALSE, PETSC_TRUE } PetscBool;
typedef int MPI_Comm;
#define PETSC_COMM_WORLD 0
#define PETSC_DECIDE (-1)
#define PETSC_DEFAULT (-2)
#define CHKERRQ(ierr) do { if (ierr) return (ierr); } while (0)

typedef struct _p_Vec *Vec;
typedef struct _p_Mat *Mat;
typedef struct _p_KSP *KSP;
typedef struct _p_PC *PC;
typedef struct _p_DM *DM;
typedef enum { NORM_1 = 0, NORM_2 = 1, NORM_FROBENIUS = 2, NORM_INFINITY = 3 } NormType;
typedef enum { NOT_SET_VALUES, INSERT_VALUES, ADD_VALUES } InsertMode;
typedef enum { DM_BOUNDARY_NONE, DM_BOUNDARY_GHOSTED, DM_BOUNDARY_MIRROR, DM_BOUNDARY_PERIODIC } DMBoundaryType;
typedef enum { DMDA_STENCIL_STAR, DMDA_STENCIL_BOX } DMDAStencilType;
typedef const char *KSPType;
typedef const char *PCType;
#define KSPCG "cg"
#define KSPFGMRES "fgmres"
#define KSPGMRES "gmres"
#define KSPCHEBYSHEV "chebyshev"
#define PCMG "mg"
#define PCJACOBI "jacobi"
#define PCSOR "sor"

/* ---- Sys (main.cc:24, :138) */
PetscErrorCode PetscInitialize(int *argc, char ***args, const char file[], const char help[]);
PetscErrorCode PetscFinalize(void);

/* ---- DMDA (TopOpt.cc:225-300; LinearElasticity.cc:60-135).  A DM with dof = 3 (or the node DM named in
 *      DMTopOptSetNodal) owns the tp_grid; element DMs only size their vectors. */
PetscErrorCode DMDACreate3d(MPI_Comm comm, DMBoundaryType bx, DMBoundaryType by, DMBoundaryType bz, DMDAStencilType st,
                            PetscInt M, PetscInt N, PetscInt P, PetscInt m, PetscInt n, PetscInt p, PetscInt dof,
                            PetscInt s, const PetscInt lx[], const PetscInt ly[], const PetscInt lz[], DM *da);
PetscErrorCode DMSetFromOptions(DM da);
PetscErrorCode DMSetUp(DM da);
PetscErrorCode DMDASetUniformCoordinates(DM da, PetscReal xmin, PetscReal xmax, PetscReal ymin, PetscReal ymax,
                                         PetscReal zmin, PetscReal zmax);
PetscErrorCode DMDAGetInfo(DM da, PetscInt *dim, PetscInt *M, PetscInt *N, PetscInt *P, PetscInt *m, PetscInt *n,
                           PetscInt *p, PetscInt *dof, PetscInt *s, DMBoundaryType *bx, DMBoundaryType *by,
                           DMBoundaryType *bz, DMDAStencilType *st);
PetscErrorCode DMCreateGlobalVector(DM da, Vec *v);
PetscErrorCode DMCreateLocalVector(DM da, Vec *v);
PetscErrorCode DMGlobalToLocalBegin(DM da, Vec g, InsertMode mode, Vec l);
PetscErrorCode DMGlobalToLocalEnd(DM da, Vec g, InsertMode mode, Vec l);
PetscErrorCode DMDestroy(DM *da);

/* ---- Vec (the calls of LinearElasticity.cc / Filter.cc / main.cc) */
PetscErrorCode VecDuplicate(Vec v, Vec *newv);
PetscErrorCode VecDestroy(Vec *v);
PetscErrorCode VecSet(Vec v, PetscScalar a);
PetscErrorCode VecCopy(Vec x, Vec y);
PetscErrorCode VecScale(Vec v, PetscScalar a);
PetscErrorCode VecAXPY(Vec y, PetscScalar a, Vec x);
PetscErrorCode VecPointwiseMult(Vec w, Vec x, Vec y);
PetscErrorCode VecPointwiseDivide(Vec w, Vec x, Vec y);
PetscErrorCode VecDot(Vec x, Vec y, PetscScalar *val);
PetscErrorCode VecNorm(Vec x, NormType type, PetscReal *val);   /* NORM_2 */
PetscErrorCode VecSum(Vec x, PetscScalar *sum);
PetscErrorCode VecGetSize(Vec x, PetscInt *n);
PetscErrorCode VecGetLocalSize(Vec x, PetscInt *n);
PetscErrorCode VecGetArray(Vec x, PetscScalar **a);       /* host mirror, copied from HBM */
PetscErrorCode VecRestoreArray(Vec x, PetscScalar **a);   /* copied back to HBM */
PetscErrorCode VecTopOptGetDevicePointer(Vec x, PetscScalar **d);   /* extension: the HBM array itself */

/* ---- Mat */
PetscErrorCode MatMult(Mat A, Vec x, Vec y);   /* elasticity: y = (N K(E) N + I - N) x; filter: y = H x */
PetscErrorCode MatDestroy(Mat *A);
/* stands for DMCreateMatrix(da_nodal,&K) + the hierarchy of SetUpSolver (LinearElasticity.cc:104-105, :551-783):
 * KE from Hex8Isoparametric (:841-998) with Poisson ratio nu, nlvls multigrid levels */
PetscErrorCode MatCreateTopOptElasticity(DM da_nodal, PetscScalar nu, PetscInt nlvls, Mat *K);
/* SetUpLoadAndBC (LinearElasticity.cc:143-176): fills N and RHS with the cantilever case and registers N */
PetscErrorCode MatTopOptCantilever(Mat K, Vec N, Vec RHS);
/* any other Dirichlet mask (1 = free, 0 = clamped), the role of MatDiagonalScale/Set in :532-538 */
PetscErrorCode MatTopOptSetDirichlet(Mat K, Vec N);
/* AssembleStiffnessMatrix (LinearElasticity.cc:487-549): E_e = Emin + xPhys^penal (Emax - Emin), Galerkin operators */
PetscErrorCode MatTopOptAssemble(Mat K, Vec xPhys, PetscScalar Emin, PetscScalar Emax, PetscScalar penal);
/* the element loop of ComputeObjectiveConstraintsSensitivities (LinearElasticity.cc:405-437) */
PetscErrorCode MatTopOptComplianceSensitivity(Mat K, Vec U, Vec xPhys, PetscScalar Emin, PetscScalar Emax,
                                              PetscScalar penal, PetscScalar volfrac, PetscScalar *fx, PetscScalar *gx,
                                              Vec dfdx, Vec dgdx);
/* Filter::SetUp (Filter.cc:290-463): H (cone filter, types 0/1) and Hs = H 1 on the element grid of da_nodes */
PetscErrorCode MatCreateTopOptFilter(DM da_nodes, PetscInt filterType, PetscScalar R, Mat *H, Vec *Hs);

/* ---- KSP / PC (LinearElasticity.cc:182-223, :617-746) */
PetscErrorCode KSPCreate(MPI_Comm comm, KSP *ksp);
PetscErrorCode KSPSetType(KSP ksp, KSPType type);            /* KSPCG; others: error 56 (PETSC_ERR_SUP) */
PetscErrorCode KSPSetTolerances(KSP ksp, PetscReal rtol, PetscReal abstol, PetscReal dtol, PetscInt maxits);
PetscErrorCode KSPSetInitialGuessNonzero(KSP ksp, PetscBool flg);
PetscErrorCode KSPSetOperators(KSP ksp, Mat A, Mat P);
PetscErrorCode KSPSetFromOptions(KSP ksp);
PetscErrorCode KSPSetUp(KSP ksp);
PetscErrorCode KSPSolve(KSP ksp, Vec b, Vec x);
PetscErrorCode KSPGetIterationNumber(KSP ksp, PetscInt *its);
PetscErrorCode KSPGetResidualNorm(KSP ksp, PetscReal *rnorm);
PetscErrorCode KSPGetPC(KSP ksp, PC *pc);
PetscErrorCode KSPDestroy(KSP *ksp);
PetscErrorCode PCSetType(PC pc, PCType type);                /* PCMG */

#ifdef __cplusplus
}
#endif
#endif

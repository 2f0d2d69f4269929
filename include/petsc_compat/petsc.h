/* petsc.h (compat) -- the subset of the PETSc 3.11 C API that TopOpt_in_PETSc's hot path is written against
 * (the calls of LinearElasticity.cc, Filter.cc and PDEFilter.cc), implemented on the MI355X
 * library (libtopopt_petsc_shim.so, host/shim/ -> libtopopt_amd.so).
 *
 * Put  -I include/petsc_compat  where a PETSc build would put  -I $PETSC_DIR/include : ALL EIGHT sources of the
 * reference (main.cc, TopOpt.cc, MMA.cc, MPIIO.cc and the three hot-path classes) compile UNCHANGED against this
 * header and the small mpi.h beside it (host/build_ref_on_shim.sh and tests/test_reference_on_shim.py do
 * exactly that, in the build container, storing nothing), and the resulting program runs the reference's
 * optimisation loop on the GPU.  Same names, argument order, ownership rules (XxxDestroy nulls the handle,
 * reference counted where the reference relies on it: PCMGSetInterpolation / KSPSetOperators retain) and error
 * convention (PetscErrorCode, 0 = success, CHKERRQ = return on non-zero).
 *
 * What is different behind the names:
 *  - one process per GPU.  host/slabrun -n R starts R ranks of the program on one node; MPI_* is a shared-memory job
 *    (host/slab_comm.h), a DMDA is split into R z-slabs (1 x 1 x R process grid, PETSc's ownership and ghost ranges),
 *    nodal Vecs are the library's slab arrays (global vector = the owned window, local vector = the whole slab), the
 *    halo exchange underneath is host-staged or, with a GPU per rank, RCCL inside the library (topopt_amd.h);
 *  - Vec data lives in HBM; VecGetArray lends a host mirror (copied down, copied back on VecRestoreArray);
 *  - Mat is never assembled.  MatSetValuesLocal is a CAPTURE: the 24x24 element matrices of
 *    AssembleStiffnessMatrix (LinearElasticity.cc:510-524) must be multiples of one matrix -- the multiplier
 *    becomes the element modulus of the matrix-free operator --, the 8x8 / 8x1 blocks of PDEFilt::MatAssemble
 *    (PDEFilter.cc:243-267) select the Helmholtz and the element-to-node operators, the 1x1 entries of
 *    Filter::SetUp (Filter.cc:417-433) give the cone radius.  Anything else: PETSC_ERR_SUP.
 *    MatDiagonalScale(K, N, N) + MatDiagonalSet(K, I - N) register the Dirichlet vector (:532-538);
 *  - KSP/PC: the object graph of SetUpSolver (:617-746) is recorded; the configuration that is SOLVED is
 *    CG + PCMG(V, Galerkin) with Chebyshev/Jacobi smoothers.  The reference hard-codes FGMRES/GMRES/SOR and, like
 *    with real PETSc, the options database overrides it (KSPSetFromOptions; level KSPs at set-up):
 *      -ksp_type cg -mg_levels_ksp_type chebyshev -mg_levels_pc_type jacobi
 *      -mg_coarse_ksp_type chebyshev -mg_coarse_pc_type jacobi
 *    (argv of PetscInitialize, $PETSC_OPTIONS, or PetscOptionsSetValue).  Without them the hard-coded configuration
 *    itself is run (tp_solver_opts::ksp_mode = 1, csrc/refksp.h: FGMRES + PCMG with GMRES/SOR level solvers, one
 *    device) -- never a silent substitution of the algorithm.
 */
#ifndef TOPOPT_PETSC_COMPAT_H
#define TOPOPT_PETSC_COMPAT_H
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int PetscErrorCode;
typedef int PetscInt;
typedef int PetscMPIInt;
typedef double PetscScalar;
typedef double PetscReal;
typedef double PetscLogDouble;
typedef enum { PETSC_FALSE, PETSC_TRUE } PetscBool;
#include <mpi.h> /* the MPI subset of this directory (one node, shared memory: host/slab_comm.h) */
#define PETSC_COMM_WORLD MPI_COMM_WORLD
#define PETSC_COMM_SELF MPI_COMM_SELF
#define PETSC_DECIDE (-1)
#define PETSC_DETERMINE PETSC_DECIDE
#define PETSC_DEFAULT (-2)
#define PETSC_MAX_PATH_LEN 4096
#define PETSC_NULL NULL
#define MPIU_SCALAR MPI_DOUBLE
#define MPIU_REAL MPI_DOUBLE
#define MPIU_INT MPI_INT
#define PETSC_ERR_SUP 56
#define PETSC_ERR_LIB 76
#define PETSC_ERR_ORDER 58
#define PETSC_ERR_ARG_OUTOFRANGE 63
#define PETSC_ERR_ARG_WRONG 62
#define PETSC_ERR_FILE_OPEN 65
#define CHKERRQ(ierr) do { if (ierr) return (ierr); } while (0)
#define SETERRQ(comm, n, s) do { fprintf(stderr, "[petsc-compat] %s\n", s); return (n); } while (0)
#define PetscMin(a, b) (((a) < (b)) ? (a) : (b))
#define PetscMax(a, b) (((a) < (b)) ? (b) : (a))
#define PetscAbsScalar(a) fabs(a)
#define PetscAbsReal(a) fabs(a)
#define PetscSqrtScalar(a) sqrt(a)
#define PetscSqrtReal(a) sqrt(a)
#define PetscPowScalar(a, b) pow(a, b)
#define PetscPowReal(a, b) pow(a, b)
#define PetscRealPart(a) (a)

typedef struct _p_PetscObject *PetscObject;
typedef struct _p_Vec *Vec;
typedef struct _p_Mat *Mat;
typedef struct _p_KSP *KSP;
typedef struct _p_PC *PC;
typedef struct _p_DM *DM;
typedef struct _p_PetscViewer *PetscViewer;
typedef struct _p_PetscRandom *PetscRandom;
typedef struct _p_ISLocalToGlobalMapping *ISLocalToGlobalMapping;
typedef struct _p_IS *IS;
typedef struct _n_PetscOptions *PetscOptions;

typedef enum { NORM_1 = 0, NORM_2 = 1, NORM_FROBENIUS = 2, NORM_INFINITY = 3 } NormType;
typedef enum { NOT_SET_VALUES, INSERT_VALUES, ADD_VALUES } InsertMode;
typedef enum { MAT_FLUSH_ASSEMBLY = 1, MAT_FINAL_ASSEMBLY = 0 } MatAssemblyType;
typedef enum { DM_BOUNDARY_NONE, DM_BOUNDARY_GHOSTED, DM_BOUNDARY_MIRROR, DM_BOUNDARY_PERIODIC } DMBoundaryType;
typedef enum { DMDA_STENCIL_STAR, DMDA_STENCIL_BOX } DMDAStencilType;
typedef enum { DMDA_ELEMENT_P1, DMDA_ELEMENT_Q1 } DMDAElementType;
typedef enum { PC_MG_MULTIPLICATIVE, PC_MG_ADDITIVE, PC_MG_FULL, PC_MG_KASKADE } PCMGType;
typedef enum { PC_MG_CYCLE_V = 1, PC_MG_CYCLE_W = 2 } PCMGCycleType;
typedef enum { PC_MG_GALERKIN_BOTH, PC_MG_GALERKIN_PMAT, PC_MG_GALERKIN_MAT, PC_MG_GALERKIN_NONE, PC_MG_GALERKIN_EXTERNAL } PCMGGalerkinType;
typedef enum { FILE_MODE_READ, FILE_MODE_WRITE, FILE_MODE_APPEND, FILE_MODE_UPDATE, FILE_MODE_APPEND_UPDATE } PetscFileMode;
typedef const char *KSPType;
typedef const char *PCType;
typedef const char *PetscRandomType;
typedef const char *PetscViewerType;
#define KSPCG "cg"
#define KSPFGMRES "fgmres"
#define KSPGMRES "gmres"
#define KSPCHEBYSHEV "chebyshev"
#define KSPRICHARDSON "richardson"
#define PCMG "mg"
#define PCJACOBI "jacobi"
#define PCSOR "sor"
#define PCGAMG "gamg"
#define PCSPAI "spai"
#define PCNONE "none"
#define PETSCRAND48 "rand48"
#define PETSCRAND "rand"
#define PETSCVIEWERASCII "ascii"
#define PETSCVIEWERBINARY "binary"

typedef struct {
    PetscInt dim, dof, sw;
    PetscInt mx, my, mz;    /* global number of grid points in each direction */
    PetscInt xs, ys, zs;    /* starting point of this processor, excluding ghosts */
    PetscInt xm, ym, zm;    /* number of grid points on this processor, excluding ghosts */
    PetscInt gxs, gys, gzs; /* starting point of this processor including ghosts */
    PetscInt gxm, gym, gzm; /* number of grid points on this processor including ghosts */
    DMBoundaryType bx, by, bz;
    DMDAStencilType st;
    DM da;
} DMDALocalInfo;

/* ---- Sys */
PetscErrorCode PetscInitialize(int *argc, char ***args, const char file[], const char help[]);
PetscErrorCode PetscFinalize(void);
PetscErrorCode PetscOptionsSetValue(PetscOptions o, const char name[], const char value[]);
PetscErrorCode PetscOptionsClearValue(PetscOptions o, const char name[]);
PetscErrorCode PetscOptionsGetInt(PetscOptions o, const char pre[], const char name[], PetscInt *v, PetscBool *set);
PetscErrorCode PetscOptionsGetReal(PetscOptions o, const char pre[], const char name[], PetscReal *v, PetscBool *set);
PetscErrorCode PetscOptionsGetBool(PetscOptions o, const char pre[], const char name[], PetscBool *v, PetscBool *set);
PetscErrorCode PetscOptionsGetString(PetscOptions o, const char pre[], const char name[], char s[], size_t len, PetscBool *set);
PetscErrorCode PetscPrintf(MPI_Comm comm, const char format[], ...);
PetscErrorCode PetscErrorPrintf(const char format[], ...);
PetscErrorCode PetscMallocCompat(size_t n, void **p);
PetscErrorCode PetscFreeCompat(void *p);
#define PetscMalloc(n, p) PetscMallocCompat((size_t)(n), (void **)(p))
#define PetscFree(p) (PetscFreeCompat((void *)(p)), (p) = 0, 0)
PetscErrorCode PetscObjectTypeCompare(PetscObject obj, const char type_name[], PetscBool *same);
PetscErrorCode PetscViewerBinaryOpen(MPI_Comm comm, const char name[], PetscFileMode mode, PetscViewer *v);
PetscErrorCode PetscViewerCreate(MPI_Comm comm, PetscViewer *v);
PetscErrorCode PetscViewerSetType(PetscViewer v, PetscViewerType type);
PetscErrorCode PetscViewerFileSetMode(PetscViewer v, PetscFileMode mode);
PetscErrorCode PetscViewerFileSetName(PetscViewer v, const char name[]);
PetscErrorCode PetscViewerASCIIPrintf(PetscViewer v, const char format[], ...);
PetscErrorCode PetscViewerDestroy(PetscViewer *v);
PetscErrorCode PetscRandomCreate(MPI_Comm comm, PetscRandom *r);
PetscErrorCode PetscRandomSetType(PetscRandom r, PetscRandomType type);
PetscErrorCode PetscRandomDestroy(PetscRandom *r);

/* ---- DM / DMDA (TopOpt.cc:225-300; LinearElasticity.cc:46-180; Filter.cc:339-372; PDEFilter.cc:28-141) */
PetscErrorCode DMDACreate3d(MPI_Comm comm, DMBoundaryType bx, DMBoundaryType by, DMBoundaryType bz, DMDAStencilType st,
                            PetscInt M, PetscInt N, PetscInt P, PetscInt m, PetscInt n, PetscInt p, PetscInt dof,
                            PetscInt s, const PetscInt lx[], const PetscInt ly[], const PetscInt lz[], DM *da);
PetscErrorCode DMSetFromOptions(DM da);
PetscErrorCode DMSetUp(DM da);
PetscErrorCode DMDASetUniformCoordinates(DM da, PetscReal xmin, PetscReal xmax, PetscReal ymin, PetscReal ymax,
                                         PetscReal zmin, PetscReal zmax);
PetscErrorCode DMDASetElementType(DM da, DMDAElementType t);
PetscErrorCode DMDAGetInfo(DM da, PetscInt *dim, PetscInt *M, PetscInt *N, PetscInt *P, PetscInt *m, PetscInt *n,
                           PetscInt *p, PetscInt *dof, PetscInt *s, DMBoundaryType *bx, DMBoundaryType *by,
                           DMBoundaryType *bz, DMDAStencilType *st);
PetscErrorCode DMDAGetCorners(DM da, PetscInt *x, PetscInt *y, PetscInt *z, PetscInt *m, PetscInt *n, PetscInt *p);
PetscErrorCode DMDAGetGhostCorners(DM da, PetscInt *x, PetscInt *y, PetscInt *z, PetscInt *m, PetscInt *n, PetscInt *p);
PetscErrorCode DMDAGetOwnershipRanges(DM da, const PetscInt *lx[], const PetscInt *ly[], const PetscInt *lz[]);
PetscErrorCode DMDAGetLocalInfo(DM da, DMDALocalInfo *info);
PetscErrorCode DMDAGetElements(DM da, PetscInt *nel, PetscInt *nen, const PetscInt *e[]);
PetscErrorCode DMDARestoreElements(DM da, PetscInt *nel, PetscInt *nen, const PetscInt *e[]);
PetscErrorCode DMGetCoordinatesLocal(DM da, Vec *c);                 /* borrowed */
PetscErrorCode DMGetLocalToGlobalMapping(DM da, ISLocalToGlobalMapping *m); /* borrowed */
PetscErrorCode DMCreateGlobalVector(DM da, Vec *v);
PetscErrorCode DMCreateLocalVector(DM da, Vec *v);
PetscErrorCode DMCreateMatrix(DM da, Mat *A);
PetscErrorCode DMGlobalToLocalBegin(DM da, Vec g, InsertMode mode, Vec l);
PetscErrorCode DMGlobalToLocalEnd(DM da, Vec g, InsertMode mode, Vec l);
PetscErrorCode DMCoarsenHierarchy(DM da, PetscInt nlevels, DM dac[]);
PetscErrorCode DMCreateInterpolation(DM dac, DM daf, Mat *P, Vec *scale);
PetscErrorCode DMDestroy(DM *da);

/* ---- Vec */
PetscErrorCode VecDuplicate(Vec v, Vec *newv);
PetscErrorCode VecDuplicateVecs(Vec v, PetscInt m, Vec *V[]);
PetscErrorCode VecDestroyVecs(PetscInt m, Vec *V[]);
PetscErrorCode VecDestroy(Vec *v);
PetscErrorCode VecSet(Vec v, PetscScalar a);
PetscErrorCode VecCopy(Vec x, Vec y);
PetscErrorCode VecScale(Vec v, PetscScalar a);
PetscErrorCode VecAXPY(Vec y, PetscScalar a, Vec x);
PetscErrorCode VecAXPBY(Vec y, PetscScalar a, PetscScalar b, Vec x);
PetscErrorCode VecPointwiseMult(Vec w, Vec x, Vec y);
PetscErrorCode VecPointwiseDivide(Vec w, Vec x, Vec y);
PetscErrorCode VecDot(Vec x, Vec y, PetscScalar *val);
PetscErrorCode VecNorm(Vec x, NormType type, PetscReal *val);
PetscErrorCode VecSum(Vec x, PetscScalar *sum);
PetscErrorCode VecMax(Vec x, PetscInt *p, PetscReal *val);
PetscErrorCode VecMin(Vec x, PetscInt *p, PetscReal *val);
PetscErrorCode VecGetSize(Vec x, PetscInt *n);
PetscErrorCode VecGetLocalSize(Vec x, PetscInt *n);
PetscErrorCode VecGetArray(Vec x, PetscScalar **a);
PetscErrorCode VecRestoreArray(Vec x, PetscScalar **a);
PetscErrorCode VecGetArrays(const Vec x[], PetscInt n, PetscScalar **a[]);
PetscErrorCode VecRestoreArrays(const Vec x[], PetscInt n, PetscScalar **a[]);
PetscErrorCode VecAXPBYPCZ(Vec z, PetscScalar alpha, PetscScalar beta, PetscScalar gamma, Vec x, Vec y);
PetscErrorCode VecSetValueLocal(Vec v, PetscInt row, PetscScalar value, InsertMode mode);
PetscErrorCode VecSetValue(Vec v, PetscInt row, PetscScalar value, InsertMode mode);
PetscErrorCode VecAssemblyBegin(Vec v);
PetscErrorCode VecAssemblyEnd(Vec v);
PetscErrorCode VecSetRandom(Vec v, PetscRandom r);
PetscErrorCode VecView(Vec v, PetscViewer viewer);
PetscErrorCode VecLoad(Vec v, PetscViewer viewer);
PetscErrorCode VecTopOptGetDevicePointer(Vec x, PetscScalar **d); /* extension: the HBM array itself */

/* ---- Mat */
PetscErrorCode MatCreateAIJ(MPI_Comm comm, PetscInt m, PetscInt n, PetscInt M, PetscInt N, PetscInt d_nz,
                            const PetscInt d_nnz[], PetscInt o_nz, const PetscInt o_nnz[], Mat *A);
PetscErrorCode MatSetLocalToGlobalMapping(Mat A, ISLocalToGlobalMapping r, ISLocalToGlobalMapping c);
PetscErrorCode MatZeroEntries(Mat A);
PetscErrorCode MatSetValuesLocal(Mat A, PetscInt nrow, const PetscInt irow[], PetscInt ncol, const PetscInt icol[],
                                 const PetscScalar y[], InsertMode addv);
PetscErrorCode MatAssemblyBegin(Mat A, MatAssemblyType t);
PetscErrorCode MatAssemblyEnd(Mat A, MatAssemblyType t);
PetscErrorCode MatDiagonalScale(Mat A, Vec l, Vec r);
PetscErrorCode MatDiagonalSet(Mat A, Vec d, InsertMode mode);
PetscErrorCode MatMult(Mat A, Vec x, Vec y);
PetscErrorCode MatMultTranspose(Mat A, Vec x, Vec y);
PetscErrorCode MatDestroy(Mat *A);

/* ---- KSP / PC */
PetscErrorCode KSPCreate(MPI_Comm comm, KSP *ksp);
PetscErrorCode KSPSetType(KSP ksp, KSPType type);
PetscErrorCode KSPGetType(KSP ksp, KSPType *type);
PetscErrorCode KSPGMRESSetRestart(KSP ksp, PetscInt restart);
PetscErrorCode KSPSetTolerances(KSP ksp, PetscReal rtol, PetscReal abstol, PetscReal dtol, PetscInt maxits);
PetscErrorCode KSPGetTolerances(KSP ksp, PetscReal *rtol, PetscReal *abstol, PetscReal *dtol, PetscInt *maxits);
PetscErrorCode KSPSetInitialGuessNonzero(KSP ksp, PetscBool flg);
PetscErrorCode KSPSetOperators(KSP ksp, Mat A, Mat P);
PetscErrorCode KSPSetFromOptions(KSP ksp);
PetscErrorCode KSPSetUp(KSP ksp);
PetscErrorCode KSPSolve(KSP ksp, Vec b, Vec x);
PetscErrorCode KSPGetIterationNumber(KSP ksp, PetscInt *its);
PetscErrorCode KSPGetResidualNorm(KSP ksp, PetscReal *rnorm);
PetscErrorCode KSPGetPC(KSP ksp, PC *pc);
PetscErrorCode KSPDestroy(KSP *ksp);
PetscErrorCode PCSetType(PC pc, PCType type);
PetscErrorCode PCGetType(PC pc, PCType *type);
PetscErrorCode PCSetReusePreconditioner(PC pc, PetscBool flag);
PetscErrorCode PCMGSetLevels(PC pc, PetscInt levels, MPI_Comm *comms);
PetscErrorCode PCMGSetType(PC pc, PCMGType form);
PetscErrorCode PCMGSetCycleType(PC pc, PCMGCycleType n);
PetscErrorCode PCMGSetGalerkin(PC pc, PCMGGalerkinType use);
PetscErrorCode PCMGSetInterpolation(PC pc, PetscInt l, Mat mat);
PetscErrorCode PCMGGetCoarseSolve(PC pc, KSP *ksp);
PetscErrorCode PCMGGetSmoother(PC pc, PetscInt l, KSP *ksp);
/* extension: the PETSc option string (numeric Chebyshev windows) of the solver behind `ksp` after KSPSetUp */
PetscErrorCode KSPTopOptGetOptionString(KSP ksp, char buf[], size_t len);

/* ---- extension calls of the thin adapter (host/shim_le.cc): the same operators without the capture detour */
PetscErrorCode MatCreateTopOptElasticity(DM da_nodal, PetscScalar nu, PetscInt nlvls, Mat *K);
PetscErrorCode MatTopOptCantilever(Mat K, Vec N, Vec RHS);
PetscErrorCode MatTopOptSetDirichlet(Mat K, Vec N);
PetscErrorCode MatTopOptAssemble(Mat K, Vec xPhys, PetscScalar Emin, PetscScalar Emax, PetscScalar penal);
PetscErrorCode MatTopOptComplianceSensitivity(Mat K, Vec U, Vec xPhys, PetscScalar Emin, PetscScalar Emax,
                                              PetscScalar penal, PetscScalar volfrac, PetscScalar *fx, PetscScalar *gx,
                                              Vec dfdx, Vec dgdx);
/* filterType 0 / 1: H (cone filter) and Hs = H 1 (Filter.cc:290-463); 2: the PDE filter as one operator,
 * MatMult(H, x, y) = T^T K_f^-1 (vol T x) (PDEFilter.cc:189-216), Hs = 1 */
PetscErrorCode MatCreateTopOptFilter(DM da_nodes, PetscInt filterType, PetscScalar R, Mat *H, Vec *Hs);

#ifdef __cplusplus
}
#endif
#endif

/*
 * topopt_amd.h -- C ABI of the MI355X-native hot path of TopOpt_in_PETSc.
 *
 * One shared library (libtopopt_amd.so, HIP, gfx950) replaces what the
 * reference does per design iteration through PETSc:
 *
 *   stiffness "assembly"  -> tp_elasticity_assemble   (LinearElasticity.cc:487-549, :198-200)
 *   KSPSolve (CG + PCMG)  -> tp_elasticity_solve      (LinearElasticity.cc:204, :617-746)
 *   MatMult(K, u)         -> tp_elasticity_apply      (PETSc MatMult on the assembled K)
 *   objective/sensitivity -> tp_elasticity_objective  (LinearElasticity.cc:363-445)
 *   density filter        -> tp_filter_*              (Filter.cc:60-204, :290-463)
 *   Helmholtz PDE filter  -> tp_pdefilter_*           (PDEFilter.cc:189-218, :243-417)
 *
 * Conventions (mirroring PetscErrorCode): every function returns int, 0 = OK,
 * nonzero = error (TP_ERR_*; hipError_t values are passed through offset by
 * TP_ERR_HIP).  No exceptions cross the boundary.  Scalars are double
 * (PetscScalar), indices int (PetscInt, 32 bit) except sizes in bytes/long.
 *
 * Memory: every `double*` argument marked [dev] is a DEVICE pointer to plain
 * contiguous doubles (no torch types).  Host code that only has host arrays
 * (VecGetArray in the reference) uses tp_malloc / tp_memcpy_* below.
 *
 * Layout (identical to the reference's DMDA natural ordering,
 * LinearElasticity.cc:819-826): node id = i + nx*(j + ny*k), dof = 3*node+c
 * (u,v,w interlaced), element id = i + ex*(j + ey*k), x fastest.
 *
 * Partitioning: z-slabs.  Rank r of R owns element layers [r*ez/R, (r+1)*ez/R)
 * and, like DMDA (a rank owns the elements whose upper corner node it owns,
 * LinearElasticity.cc:802-814), the node planes (r*ez/R, (r+1)*ez/R]; rank 0
 * also owns plane 0.  A rank's *local* node array stores planes
 * r*ez/R .. (r+1)*ez/R+1 (one ghost plane either side where a neighbour
 * exists); local element arrays store the own layers only.  With R = 1 local
 * == global.  tp_grid_local_* report the local sizes.
 *
 * Threading: one host thread per context; all work is enqueued on the
 * hipStream_t given at creation (pass the framework's current stream).
 */
#ifndef TOPOPT_AMD_H
#define TOPOPT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TP_OK 0
#define TP_ERR_ARG 1        /* bad argument / mesh not coarsenable (TopOpt.cc:183-201) */
#define TP_ERR_STATE 2      /* call order violated (e.g. solve before assemble) */
#define TP_ERR_DIVERGED 3   /* KSP_DIVERGED_DTOL analogue */
#define TP_ERR_COMM 4       /* communication callback failed */
#define TP_ERR_HIP 1000     /* TP_ERR_HIP + hipError_t */

#define TP_MAX_LEVELS 10

/* ---- communication hooks (z-slab halo + reductions) -------------------- */
/* tp_comm is the set of slab-exchange operations the library calls: a staged neighbour exchange, an all-reduce
 * of up to 16 doubles and an all-gather on staging buffers the host framework owns, plus two optional in-place
 * forms (zero-copy halo of contiguous planes, in-place all-reduce).  All of them are ordered on the stream the grid
 * was created with.  Two implementations exist: the host framework's hooks (torch.distributed with the nccl(=RCCL)
 * backend, gloo in the CPU / one-GPU tests, MPI in a C++ host) and the library's own RCCL path, which replaces the
 * hooks after tp_grid_use_rccl (RCCL is dlopen'ed from the path the host passes, never linked).               */
typedef struct tp_comm {
    void *user;
    double *send_lo, *send_hi, *recv_lo, *recv_hi; /* [dev] staging, cap doubles each */
    double *red;                                   /* [dev] >= 16 doubles, all-reduce scratch */
    long cap;
    /* send send_lo[0..n) to rank-1 and send_hi[0..n) to rank+1, receive recv_lo from
     * rank-1 and recv_hi from rank+1 (absent neighbours are skipped). */
    int (*exchange)(void *user, long n);
    /* in-place sum over ranks of red[0..n) */
    int (*allreduce_sum)(void *user, int n);
    /* gather[r*n .. (r+1)*n) <- rank r's send_lo[0..n), for every rank (n <= cap) */
    double *gather;                                /* [dev] nranks * cap doubles */
    int (*allgather)(void *user, long n);
    /* optional zero-copy halo of contiguous planes (may be NULL -> staged `exchange` is used):
     * to_lo[0..n) -> rank-1, to_hi[0..n) -> rank+1, from_lo <- rank-1, from_hi <- rank+1, all [dev].
     * Returns 0, or 2 = "cannot address these pointers, nothing was sent": the library then uses `exchange`
     * from now on (same message sizes and order, so ranks may differ in their choice); else error. */
    int (*exchange_direct)(void *user, const double *to_lo, double *from_lo, const double *to_hi, double *from_hi,
                           long n);
    /* optional: in-place sum over ranks of p[0..n), p any [dev] address (NULL -> staged through `red`) */
    int (*allreduce_inplace)(void *user, double *p, int n);
    /* optional: issue the FOLLOWING operations on `stream` (a hipStream_t) instead of the grid's stream, until the
     * next call (NULL restores the grid's stream).  With it and exchange_direct the library overlaps the halo of a
     * smoothing step with the step's interior planes on a second stream (DMGlobalToLocalBegin ... End,
     * LinearElasticity.cc:249-250); NULL -> every halo is exchanged on the grid's stream before its consumer. */
    void (*set_stream)(void *user, void *stream);
} tp_comm;

/* ---- grid / partition --------------------------------------------------- */
typedef struct tp_grid_opts {
    int nx, ny, nz;       /* GLOBAL node counts, TopOpt.cc:106-108 (-nx -ny -nz) */
    double hx, hy, hz;    /* element edge lengths */
    int rank, nranks;     /* z-slab partition */
    int device;           /* HIP device ordinal */
    void *stream;         /* hipStream_t; NULL = the null stream */
    const tp_comm *comm;  /* NULL when nranks == 1 */
} tp_grid_opts;

typedef struct tp_grid tp_grid;
int tp_grid_create(tp_grid **g, const tp_grid_opts *o);
int tp_grid_destroy(tp_grid *g);
long tp_grid_local_nodes(const tp_grid *g);     /* nodes in the local array incl. ghost planes */
long tp_grid_local_elems(const tp_grid *g);     /* own elements */
long tp_grid_owned_node_offset(const tp_grid *g); /* first owned node in the local array */
long tp_grid_owned_nodes(const tp_grid *g);
int tp_grid_node_z0(const tp_grid *g);          /* global z index of local node plane 0 */
/* refresh the ghost node planes of a slab-local nodal array with `dof` values per node (DMGlobalToLocalBegin/End,
 * LinearElasticity.cc:249-250); a no-op on one rank */
int tp_grid_halo_nodes(tp_grid *g, double *v, int dof);
int tp_grid_elem_z0(const tp_grid *g);          /* global z index of local element layer 0 */

/* ---- optional: the slab exchange issued directly to RCCL -------------------------------------
 * The reference exchanges ghost layers with MPI inside PETSc (DMGlobalToLocal, VecDot; e.g.
 * LinearElasticity.cc:249-250).  By default the library calls the tp_comm hooks of the host framework; on one
 * node it can instead issue grouped ncclSend/ncclRecv, ncclAllReduce and ncclAllGather itself on the solver's
 * stream (xGMI, no host round trip per halo).  RCCL is not linked: pass the path of the librccl.so the process
 * already uses.  tp_grid_use_rccl is collective over the ranks of the grid (rank 0 creates the id, the host
 * broadcasts the 128 bytes); on failure the tp_comm hooks given at creation stay in place. */
int tp_rccl_load(const char *librccl_path);
int tp_rccl_unique_id(void *id128);
int tp_grid_use_rccl(tp_grid *g, const void *id128);
/* the same with a second unique id: the neighbour exchanges (halo stream) get a communicator of their own, so that RCCL
 * does not order them behind the all-reduces of the solver's stream (one communicator = one queue) */
int tp_grid_use_rccl2(tp_grid *g, const void *id128, const void *id128_halo);
int tp_grid_comm_stats(const tp_grid *g, long *exchanges, long *reductions);   /* RCCL path only, else zeros */
/* ranks the RCCL communicator itself reports (ncclCommCount; 0: no RCCL path, -1: unknown), second communicator in use */
int tp_grid_comm_info(const tp_grid *g, int *rccl_ranks, int *two_communicators);
/* The roofline kernel timed where it runs (bench.py): with on = 1 every launch of the fine level's fused operator +
 * Chebyshev step is bracketed by a pair of HIP events on the grid's stream; the read waits for the stream, returns
 * the summed elapsed time and the number of launches, and clears the list. */
/* Timing of the slab communication where it runs (N > 1): per kind -- 0 blocking halo exchange, 1 halo exchange overlapped on
 * the second stream, 2 all-reduce, 3 all-gather of the replicated coarse levels -- hook calls, host wall time inside the hooks and
 * device time between HIP event pairs around them.  What the reference spends in DMGlobalToLocalBegin/End
 * (/root/reference/LinearElasticity.cc:249-250) and MPI_Allreduce (:283, :429). */
int tp_grid_comm_timer(tp_grid *g, int on);
int tp_grid_comm_timer_read(tp_grid *g, long calls[4], double host_ms[4], double device_ms[4]);
int tp_grid_kernel_timer(tp_grid *g, int on);
int tp_grid_kernel_timer_read(tp_grid *g, double *total_ms, long *launches);
/* the same plus the algorithmic bytes of exactly the timed launches */
int tp_grid_kernel_timer_read2(tp_grid *g, double *total_ms, long *launches, double *alg_bytes);
/* halos that travelled on the second stream, overlapped with the interior planes of their producer (0 if the hooks
 * lack set_stream / exchange_direct, or with TP_OVERLAP=0) */
long tp_grid_overlapped_halos(const tp_grid *g);
/* back to the tp_comm hooks given at creation (destroys the library's communicator) */
int tp_grid_drop_rccl(tp_grid *g);
/* collective: rank-tagged buffers through the grid's CURRENT hooks (staged and in-place exchange, reductions,
 * all-gather); *ok = 1 if this rank received exactly its neighbours' data */
int tp_grid_comm_selfcheck(tp_grid *g, int *ok);
/* Stress test of the reductions that are finished inside the producing kernel (csrc/common.h, reduce_tail): `reps` dot
 * products of two generated vectors of n doubles, back to back, each also computed by the two-launch form (partial sums,
 * then k_reduce_final); *mismatches = how many differ in any bit.  (TP_NO_REDUCE_TAIL=1 switches the in-kernel form off
 * library-wide; the test then compares the two-launch form with itself.) */
int tp_grid_reduction_selftest(tp_grid *g, long n, int reps, int *mismatches);
/* one-rank loop-back check of the RCCL call sequence (the rank is its own lower and upper neighbour):
 * returns TP_OK and the largest deviation in *max_err */
int tp_rccl_selftest(int device, void *stream, long n, double *max_err);

/* ---- device memory helpers (for hosts without a GPU framework) ---------- */
int tp_set_device(int device);             /* hipSetDevice: before tp_malloc of the tp_comm staging buffers */
int tp_malloc(void **p, size_t bytes);
int tp_free(void *p);
int tp_memcpy_h2d(void *dst, const void *src, size_t bytes);
int tp_memcpy_d2h(void *dst, const void *src, size_t bytes);
int tp_sync(const tp_grid *g);

/* ---- linear elasticity -------------------------------------------------- */
typedef struct tp_solver_opts {
    int nlvls;          /* MG levels, LinearElasticity.cc:23 (-nlvls, default 4) */
    double nu;          /* Poisson ratio, :22 (-nu) */
    double rtol, atol, dtol; /* KSPSetTolerances, :621-623 */
    int max_it;         /* :625 */
    int nsmooth;        /* smoother iterations per sweep, :635 */
    int ncoarse;        /* coarse-solve iterations, :631 */
    double cheb_lo, cheb_hi; /* Chebyshev window as fractions of the eigenvalue estimate */
    int nlanczos;       /* Lanczos steps for the eigenvalue estimates */
    int fine_eig;       /* fine-level estimate: 0 = rigorous element bound (free), 1 = Lanczos like the coarse levels */
    /* 0: CG + V-cycle with Chebyshev/Jacobi smoothers -- the fast path (the option string of SURVEY 8(d));
     * 1: the configuration the reference hard-codes, run as written (a correctness mode, ONE device, ~1000 launches per
     *    Gauss-Seidel sweep): FGMRES(restart) + V-cycle whose smoothers are GMRES(nsmooth) for nsmooth iterations and whose
     *    coarse solve is GMRES(coarse_restart), at most ncoarse iterations to coarse_rtol on the preconditioned residual
     *    (LinearElasticity.cc:620-746, PDEFilter.cc:276-378; csrc/refksp.h) */
    int ksp_mode;
    int restart;         /* KSPGMRESSetRestart of the outer FGMRES, :624 (100) */
    int smooth_pc;       /* PC of the level smoothers: 0 PCJACOBI, 1 PCSOR (one local symmetric sweep, omega 1), :745 */
    int coarse_pc;       /* PC of the coarse solve, :731 */
    int coarse_restart;  /* :632 (30) */
    double coarse_rtol;  /* :628 (1e-8) */
    /* ksp_mode 0, coarsest level: 0 = Chebyshev run of ncoarse steps on [smallest Ritz value, cheb_hi * largest];
     * 1 = exact solve (banded Cholesky + explicit triangular inverse, csrc/coarse_direct.h) where the level has at most
     * 4096 rows on one rank (or is replicated) -- closer to the reference's coarse KSP to rtol 1e-8 (:628-632) than a
     * fixed polynomial; larger or distributed coarsest grids fall back to 0, and so do grids of <= 448 rows, whose
     * Chebyshev run stays inside one workgroup (2 = exact also there) */
    int coarse_direct;
} tp_solver_opts;
void tp_solver_default_opts(tp_solver_opts *o);
/* The option structs grow at their END from round to round.  A host built against an older header would have the library
 * read past its struct: hosts compare these two numbers with their own header at load time (the Python host in lib.py,
 * the C++ hosts in host/topopt_host.h) and refuse to run on a mismatch. */
#define TP_ABI_VERSION 4
int tp_abi_version(void);                 /* the library's TP_ABI_VERSION */
unsigned long tp_solver_opts_size(void);  /* the library's sizeof(tp_solver_opts) */

typedef struct tp_elasticity tp_elasticity;
/* LinearElasticity::LinearElasticity + SetUpLoadAndBC (LinearElasticity.cc:12-180):
 * computes KE (Hex8Isoparametric, :841-998).  N [dev, local nodes*3, 1 = free /
 * 0 = clamped] and RHS [dev, local nodes*3] are the Dirichlet and load vectors;
 * tp_elasticity_cantilever fills them with the reference's load case. */
int tp_elasticity_create(tp_elasticity **e, tp_grid *g, const tp_solver_opts *o);
/* the same with the caller's element matrix (row-major 24x24, reference corner order) instead of the built-in
 * Hex8Isoparametric: what a host that "assembles" with MatSetValuesLocal has computed itself (:118-123, :519-524) */
int tp_elasticity_create_ke(tp_elasticity **e, tp_grid *g, const tp_solver_opts *o, const double *ke_host_576);
int tp_elasticity_destroy(tp_elasticity *e);
int tp_elasticity_get_ke(const tp_elasticity *e, double *ke_host_576);
/* The element matrix the fine-level kernels apply INSIDE THE PRECONDITIONER (smoother, V-cycle residual): KE in its packed
 * Walsh-Hadamard block form, 36 values -- the 33 structural ones and KE's three translation residues; it differs from KE by
 * less than one unit in the last place of KE's largest entry -- as a double-double pair hi + lo (host arrays of 576).
 * Test/diagnostic entry point: the parity checks hand it to the extended-precision arbiter (the 80-bit rebuild of the CPU checker). */
int tp_elasticity_get_ke_effective(const tp_elasticity *e, double *hi_host_576, double *lo_host_576);
/* The element matrix of the KRYLOV operator -- A p and the initial residual of CG (KSPSolve,
 * /root/reference/LinearElasticity.cc:204; tp_elasticity_apply_krylov): the packed form plus the translation mode's
 * column and row of T KE T / 64 exactly as KE has them (132 more values).  On a displacement field whose translation
 * dominates its strain (any iterate of the state solve) its action is KE's to rounding; the residual history follows the
 * reference's KE to 1e-12.  Same double-double convention. */
int tp_elasticity_get_ke_krylov(const tp_elasticity *e, double *hi_host_576, double *lo_host_576);
int tp_elasticity_cantilever(tp_elasticity *e, double *N, double *RHS);      /* :143-171 */
int tp_elasticity_set_bc(tp_elasticity *e, const double *N);                /* N [dev] */
/* AssembleStiffnessMatrix + KSPSetOperators/KSPSetUp (:487-549, :198-200):
 * E = Emin + x^p (Emax-Emin), Galerkin coarse operators, Jacobi diagonals,
 * Chebyshev windows.  xPhys [dev, own elements].  RHS is NOT modified here;
 * tp_elasticity_solve multiplies the load by N as :542 does. */
int tp_elasticity_assemble(tp_elasticity *e, const double *xPhys, double Emin, double Emax, double penal);
/* MatMult with the operator  N K(x) N + (I - N).  u, y [dev, local nodes*3];
 * ghost planes of u are refreshed internally, y is valid on owned planes. */
int tp_elasticity_apply(tp_elasticity *e, const double *u, double *y);
/* The same product with the operator the Krylov method multiplies with inside tp_elasticity_solve (KE_krylov, see
 * tp_elasticity_get_ke_krylov): KE's action to rounding on fields whose translation dominates their strain.
 * tp_elasticity_apply itself applies the packed form (tp_elasticity_get_ke_effective: within 5e-16 max|KE| of KE entrywise). */
int tp_elasticity_apply_krylov(tp_elasticity *e, const double *u, double *y);
/* KSPSolve(ksp, RHS, U) with a warm start from U (:204, :647).  U, RHS [dev,
 * local nodes*3].  its / rnorm as KSPGetIterationNumber / KSPGetResidualNorm
 * (:212-213).  hist (host, may be NULL) receives ||b - A x_k|| for k = 0..its,
 * at most hist_cap entries. */
int tp_elasticity_solve(tp_elasticity *e, const double *RHS, double *U, int *its, double *rnorm, double *bnorm,
                        double *hist, int hist_cap);
/* ComputeObjectiveConstraintsSensitivities minus the solve (:377-437):
 * fx = sum E_e u^T KE u, gx = sum x / n - volfrac, dfdx, dgdx = 1/n.
 * U [dev, local nodes*3], xPhys, dfdx, dgdx [dev, own elements] (dgdx may be NULL). */
int tp_elasticity_objective(tp_elasticity *e, const double *U, const double *xPhys, double Emin, double Emax,
                            double penal, double volfrac, double *fx, double *gx, double *dfdx, double *dgdx);
/* The reference's split forms.  ComputeObjectiveConstraints minus the solve (/root/reference/LinearElasticity.cc:237-294):
 * fx and gx of the state U, no sensitivities written. */
int tp_elasticity_objective_only(tp_elasticity *e, const double *U, const double *xPhys, double Emin, double Emax,
                                 double penal, double volfrac, double *fx, double *gx);
/* ComputeSensitivities (/root/reference/LinearElasticity.cc:299-361): dfdx = -p x^(p-1) (Emax - Emin) u^T KE u and
 * dgdx = 1/n (may be NULL) of the state U as it is -- no solve, no reduction, no host synchronisation. */
int tp_elasticity_sensitivities(tp_elasticity *e, const double *U, const double *xPhys, double Emin, double Emax,
                                double penal, double *dfdx, double *dgdx);
/* introspection for parity tests */
/* KSPSetTolerances (LinearElasticity.cc:646); a negative value keeps the current one (PETSC_DEFAULT) */
int tp_elasticity_set_tolerances(tp_elasticity *le, double rtol, double atol, double dtol, int max_it);
/* The parity solver as a literal PETSc 3.11 option string with the numeric per-level Chebyshev windows of the last
 * tp_elasticity_assemble (KSPSetFromOptions, LinearElasticity.cc:659; -mg_levels_N_ksp_chebyshev_eigenvalues a,b):
 * someone with a PETSc build can paste it on the reference's command line and compare the residual history.
 * Returns the length of the full string (buf receives at most cap-1 characters), or -1 before the first assembly. */
int tp_elasticity_petsc_options(const tp_elasticity *e, char *buf, size_t cap);
/* PCMGSetCycleType / PCMGSetCycleTypeOnLevel: cycles[l] cycles of level l + 1 per visit of level l (l = 0 finest, 1 = V,
 * 2 = W; PETSc's form: same right-hand side, the next cycle starts from the previous one's iterate); one cycle into the
 * coarsest level whatever is set */
int tp_elasticity_set_cycles(tp_elasticity *e, const int *cycles, int n);
int tp_elasticity_level_count(const tp_elasticity *e);
long tp_elasticity_level_nodes(const tp_elasticity *e, int level);
double tp_elasticity_level_lambda(const tp_elasticity *e, int level);
/* lower end of the Chebyshev window of the coarsest level (smallest Ritz value of its 40-step Lanczos run; the reference
 * has GMRES there and needs no window: /root/reference/src/LinearElasticity.cc:760-790); 0 on the other levels */
double tp_elasticity_level_lambda_min(const tp_elasticity *e, int level);
/* rows of the coarsest level if the last assembly factored it for the exact coarse solve (tp_solver_opts.coarse_direct),
 * 0 if the Chebyshev run is in use (option off, level too large or distributed) */
int tp_elasticity_coarse_direct_active(const tp_elasticity *e);
/* Process-wide state of the recovery from one-XCD persistent kernels that gave up (their workgroups not co-resident: a
 * shared device): how many recoveries so far, whether the one-XCD forms are off, whether the coarse factorisation is no
 * longer deferred into the head of the solve (the softer first answer when only that chain gave up).  No reference
 * counterpart: PETSc's KSPSolve (LinearElasticity.cc:204) has no persistent kernels. */
int tp_xcd_status(int *giveups, int *xcd_forms_off, int *defer_off);
int tp_elasticity_level_apply(tp_elasticity *e, int level, const double *u, double *y); /* [dev, level local dofs] */
int tp_elasticity_level_diag(tp_elasticity *e, int level, double *d);
int tp_elasticity_precond(tp_elasticity *e, const double *r, double *z); /* one V-cycle */
/* ksp_mode 1 level by level (tests): z = M^-1 r with PCJACOBI (pc 0) or PCSOR (pc 1) on a level; the level's left-
 * preconditioned GMRES(m) for at most `its` iterations on x (rtol < 0: no convergence test, as PCMG runs its smoothers) */
int tp_elasticity_level_pc(tp_elasticity *e, int level, int pc, const double *r, double *z);
int tp_elasticity_level_gmres(tp_elasticity *e, int level, int pc, int m, int its, double rtol, const double *b, double *x,
                              int zero_guess, int *its_done);
/* k Chebyshev-Jacobi steps on a level (the fused operator+update kernel): x <- smooth(b, x) */
int tp_elasticity_smooth(tp_elasticity *e, int level, const double *b, double *x, int k, int zero_guess);
int tp_elasticity_restrict(tp_elasticity *e, int level, const double *rf, double *rc);
int tp_elasticity_prolong_add(tp_elasticity *e, int level, const double *xc, double *xf);
/* bytes moved / flops of the last call, by the algorithmic model of DESIGN.md */
int tp_elasticity_last_stats(const tp_elasticity *e, double *alg_bytes, double *flops, long *kernel_launches);

/* ---- density / sensitivity filter (Filter.cc) --------------------------- */
typedef struct tp_filter tp_filter;
/* Filter::Filter + SetUp (Filter.cc:25-38, :290-463).  filterType 0 = sensitivity,
 * 1 = density, 2 = PDE (Helmholtz), other = none; rmin is an absolute length. */
int tp_filter_create(tp_filter **f, tp_grid *g, int filterType, double rmin, const tp_solver_opts *pde_opts);
/* y = H x, the un-normalised cone filter: MatMult(H, x, y) of Filter.cc:68, :173, :181 (types 0, 1) */
int tp_filter_mult_h(tp_filter *f, const double *x, double *y);
int tp_filter_destroy(tp_filter *f);
int tp_filter_stencil_width(const tp_filter *f);        /* ElemConn, Filter.cc:326 */
int tp_filter_get_hs(tp_filter *f, double *Hs);          /* [dev, own elements] */
/* PDE filter (type 2): the 8x8 Helmholtz element matrix KF of PDEFilt::PDEFilterMatrix (PDEFilter.cc:472-576), host */
int tp_filter_get_kf(const tp_filter *f, double *kf_host_64);
/* Filter::FilterProject (:60-117): x -> xTilde -> xPhys [dev, own elements] */
int tp_filter_project(tp_filter *f, const double *x, double *xTilde, double *xPhys, int projectionFilter, double beta,
                      double eta);
/* Filter::Gradients (:120-204): dfdx and dgdx[0..m) filtered in place */
int tp_filter_gradients(tp_filter *f, const double *x, const double *xTilde, double *dfdx, int m, double **dgdx,
                        int projectionFilter, double beta, double eta);
int tp_filter_mnd(tp_filter *f, const double *x, double *mnd);   /* GetMND, :206-225 */
int tp_filter_last_pde_its(const tp_filter *f, int *its, double *rnorm);
/* PDE filter (type 2), the operators of PDEFilt::FilterProject one by one (PDEFilter.cc:198-210): T x (element ->
 * node, the caller scales by the element volume), the Helmholtz solve K_f u = rhs (warm start from u), T^T u, and
 * K_f u itself.  Nodal arrays: [dev, local nodes]; element arrays: [dev, own elements]. */
int tp_pdefilter_elem_to_node(tp_filter *f, const double *x_elem, double *rhs_nodal);
int tp_pdefilter_solve(tp_filter *f, const double *rhs_nodal, double *u_nodal);
int tp_pdefilter_node_to_elem(tp_filter *f, const double *u_nodal, double *x_elem);
int tp_pdefilter_apply(tp_filter *f, const double *u_nodal, double *y_nodal);

/* ---- MMA optimizer step on the device (MMA.cc) ------------ */
typedef struct tp_mma tp_mma;
/* MMA::MMA(n, m, x) (MMA.cc:108-190): a = 0, c = 1000, d = 0, asymptote factors 0.5 / 0.7 / 1.2.
 * n_local = own design variables of this rank, n_global = all of them; x [dev, n_local]. */
int tp_mma_create(tp_mma **mma, tp_grid *g, long n_local, long n_global, int m, const double *x);
int tp_mma_destroy(tp_mma *mma);
/* SetOuterMovelimit (MMA.cc:386-405) */
int tp_mma_set_outer_movelimit(tp_mma *mma, double Xmin, double Xmax, double movlim, const double *x, double *xmin,
                               double *xmax);
/* Update (MMA.cc:499-518): x is overwritten by the new design.  gx: host array of m constraint
 * values; dgdx: host array of m DEVICE pointers.  inner_its (may be NULL): Newton steps taken. */
int tp_mma_update(tp_mma *mma, double *x, const double *dfdx, const double *gx, const double *const *dgdx,
                  const double *xmin, const double *xmax, int *inner_its);
/* DesignChange (MMA.cc:407-426): ch = max |x - xold| over all ranks, then xold <- x */
int tp_mma_design_change(tp_mma *mma, const double *x, double *xold, double *ch);
int tp_mma_get_state(const tp_mma *mma, double *lam, double *z, int *k);
/* MMA::Restart (MMA.cc:319-360): copy out the iterates/asymptotes a restart needs (device arrays, n_local each);
 * restart_set = the restart constructor (MMA.cc:22-106): continue at outer iteration k with these. */
int tp_mma_restart_get(const tp_mma *mma, double *xo1, double *xo2, double *U, double *L);
int tp_mma_restart_set(tp_mma *mma, int k, const double *xo1, const double *xo2, const double *U, const double *L);

/* ---- streaming helpers used by the driver (main.cc:68-73, TopOpt.cc) ----- */
int tp_vec_scale(tp_grid *g, double *x, double a, long n);
int tp_vec_set(tp_grid *g, double *x, double a, long n);
/* BLAS-1 surface behind the PETSc-named adapter (petsc_shim.h): VecAXPY/VecAXPBY, VecPointwiseMult/Divide,
 * VecDot/VecSum (n = the caller's OWNED range; the result is summed over the ranks of the grid) */
int tp_vec_axpby(tp_grid *g, double *y, double a, const double *x, double b, long n);
int tp_vec_pointwise(tp_grid *g, double *w, const double *x, const double *y, int divide, long n);
int tp_vec_dot(tp_grid *g, const double *x, const double *y /* NULL: sum of x */, long n, double *out);
/* synthetic density of SURVEY.md 8(d), indexed by GLOBAL element id */
int tp_synth_density(tp_grid *g, double *x, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif /* TOPOPT_AMD_H */

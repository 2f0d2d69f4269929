/* mpi.h (compat) -- the subset of MPI that TopOpt_in_PETSc touches outside PETSc: rank / size, barriers, reductions and
 * gathers over the ranks of ONE node (a shared-memory job started by host/slabrun: host/slab_comm.h; no MPI library is
 * involved), timers, and the MPI-IO calls of MPIIO.cc (file views with byte displacement and vector filetypes; every
 * rank writes through its own view into the same file).  One process per GPU; the slab exchange of the MI355X library
 * itself is tp_comm / RCCL (topopt_amd.h). */
#ifndef TOPOPT_MPI_COMPAT_H
#define TOPOPT_MPI_COMPAT_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Op;
typedef int MPI_Info;
typedef long long MPI_Offset;
typedef struct _mpi_compat_file *MPI_File;
typedef struct { int count; } MPI_Status;
#define MPI_COMM_WORLD 0
#define MPI_COMM_SELF 1
#define MPI_INFO_NULL 0
#define MPI_STATUS_IGNORE ((MPI_Status *)0)
#define MPI_SUCCESS 0
#define MPI_CHAR 3
#define MPI_DOUBLE 1
#define MPI_INT 2
#define MPI_FLOAT 4
#define MPI_UNSIGNED_LONG 5
#define MPI_LONG 6
#define MPI_SUM 1
#define MPI_MAX 2
#define MPI_MIN 3
#define MPI_MODE_CREATE 1
#define MPI_MODE_WRONLY 4
#define MPI_MODE_RDONLY 2
#define MPI_MODE_RDWR 8
int MPI_Init(int *argc, char ***argv);
int MPI_Finalize(void);
int MPI_Abort(MPI_Comm comm, int code);
int MPI_Comm_rank(MPI_Comm comm, int *rank);
int MPI_Comm_size(MPI_Comm comm, int *size);
int MPI_Barrier(MPI_Comm comm);
double MPI_Wtime(void);
int MPI_Allreduce(const void *sendbuf, void *recvbuf, int count, MPI_Datatype t, MPI_Op op, MPI_Comm comm);
int MPI_Allgather(const void *sendbuf, int sendcount, MPI_Datatype st, void *recvbuf, int recvcount, MPI_Datatype rt, MPI_Comm comm);
int MPI_Type_size(MPI_Datatype t, int *size);
int MPI_Type_vector(int count, int blocklength, int stride, MPI_Datatype oldtype, MPI_Datatype *newtype);
int MPI_Type_commit(MPI_Datatype *t);
int MPI_Type_free(MPI_Datatype *t);
int MPI_File_open(MPI_Comm comm, const char *filename, int amode, MPI_Info info, MPI_File *fh);
int MPI_File_close(MPI_File *fh);
int MPI_File_delete(const char *filename, MPI_Info info);
int MPI_File_set_view(MPI_File fh, MPI_Offset disp, MPI_Datatype etype, MPI_Datatype filetype, const char *datarep, MPI_Info info);
int MPI_File_write(MPI_File fh, const void *buf, int count, MPI_Datatype t, MPI_Status *status);
int MPI_File_write_all(MPI_File fh, const void *buf, int count, MPI_Datatype t, MPI_Status *status);
#ifdef __cplusplus
}
#endif
#endif

// shim/sys.cc -- PETSc-named surface of include/petsc_compat/petsc.h, part "sys" (see shim/internal.h)
#include "internal.h"

extern "C" {

// =============================================================================================== Sys
PetscErrorCode PetscInitialize(int *argc, char ***args, const char[], const char[]) {
    if (argc && args)
        for (int i = 1; i < *argc; i++) {
            const char *a = (*args)[i];
            if (a[0] == '-' && a[1] && !(a[1] >= '0' && a[1] <= '9')) {
                const bool val = i + 1 < *argc && !((*args)[i + 1][0] == '-' && (*args)[i + 1][1] && !((*args)[i + 1][1] >= '0' && (*args)[i + 1][1] <= '9') && (*args)[i + 1][1] != '.');
                opts()[a + 1] = val ? (*args)[i + 1] : "";
                if (val) i++;
            }
        }
    return 0;
}
PetscErrorCode PetscFinalize(void) { return 0; }
PetscErrorCode PetscOptionsSetValue(PetscOptions, const char name[], const char value[]) {
    opts()[name[0] == '-' ? name + 1 : name] = value ? value : "";
    return 0;
}
PetscErrorCode PetscOptionsClearValue(PetscOptions, const char name[]) {
    opts().erase(name[0] == '-' ? name + 1 : name);
    return 0;
}
PetscErrorCode PetscOptionsGetInt(PetscOptions, const char pre[], const char name[], PetscInt *v, PetscBool *set) {
    const std::string *s = opt_find(pre, name);
    if (set) *set = s ? PETSC_TRUE : PETSC_FALSE;
    if (s) *v = atoi(s->c_str());
    return 0;
}
PetscErrorCode PetscOptionsGetReal(PetscOptions, const char pre[], const char name[], PetscReal *v, PetscBool *set) {
    const std::string *s = opt_find(pre, name);
    if (set) *set = s ? PETSC_TRUE : PETSC_FALSE;
    if (s) *v = atof(s->c_str());
    return 0;
}
PetscErrorCode PetscOptionsGetBool(PetscOptions, const char pre[], const char name[], PetscBool *v, PetscBool *set) {
    const std::string *s = opt_find(pre, name);
    if (set) *set = s ? PETSC_TRUE : PETSC_FALSE;
    if (s) *v = (s->empty() || *s == "1" || *s == "true" || *s == "yes" || *s == "TRUE") ? PETSC_TRUE : PETSC_FALSE;
    return 0;
}
PetscErrorCode PetscOptionsGetString(PetscOptions, const char pre[], const char name[], char str[], size_t len, PetscBool *set) {
    const std::string *s = opt_find(pre, name);
    if (set) *set = s ? PETSC_TRUE : PETSC_FALSE;
    if (s && len) {
        strncpy(str, s->c_str(), len - 1);
        str[len - 1] = 0;
    }
    return 0;
}
PetscErrorCode PetscPrintf(MPI_Comm comm, const char format[], ...) {
    if (comm == MPI_COMM_WORLD && job_rank() != 0) return 0;  // the first rank of the communicator prints
    va_list ap;
    va_start(ap, format);
    vprintf(format, ap);
    va_end(ap);
    fflush(stdout);
    return 0;
}
PetscErrorCode PetscErrorPrintf(const char format[], ...) {
    va_list ap;
    va_start(ap, format);
    vfprintf(stderr, format, ap);
    va_end(ap);
    return 0;
}
PetscErrorCode PetscMallocCompat(size_t n, void **p) {
    *p = malloc(n ? n : 1);
    return *p ? 0 : 55;
}
PetscErrorCode PetscFreeCompat(void *p) {
    free(p);
    return 0;
}
PetscErrorCode PetscObjectTypeCompare(PetscObject obj, const char type_name[], PetscBool *same) {
    const Hdr *h = (const Hdr *)obj;
    const char *t = h ? h->type_name : nullptr;
    if (h && h->classid == CLS_PC) t = ((PC)obj)->type.c_str();
    if (h && h->classid == CLS_KSP) t = ((KSP)obj)->type.c_str();
    *same = (t && type_name && strcmp(t, type_name) == 0) ? PETSC_TRUE : PETSC_FALSE;
    return 0;
}
static int mpi_esize(MPI_Datatype t) {
    switch (t) {
    case MPI_CHAR: return 1;
    case MPI_INT: case MPI_FLOAT: return 4;
    default: return 8;
    }
}
struct VecType {
    int count, block, stride, esize;
};
static std::vector<VecType> &vec_types() {
    static std::vector<VecType> v;
    return v;
}
// the element types of the reference's reductions as doubles and back (counts stay far below 2^53)
static double mpi_load(const void *p, int i, MPI_Datatype t) {
    switch (t) {
    case MPI_INT: return (double)((const int *)p)[i];
    case MPI_FLOAT: return (double)((const float *)p)[i];
    case MPI_UNSIGNED_LONG: return (double)((const unsigned long *)p)[i];
    case MPI_LONG: return (double)((const long *)p)[i];
    case MPI_CHAR: return (double)((const char *)p)[i];
    default: return ((const double *)p)[i];
    }
}
static void mpi_store(void *p, int i, MPI_Datatype t, double v) {
    switch (t) {
    case MPI_INT: ((int *)p)[i] = (int)v; break;
    case MPI_FLOAT: ((float *)p)[i] = (float)v; break;
    case MPI_UNSIGNED_LONG: ((unsigned long *)p)[i] = (unsigned long)v; break;
    case MPI_LONG: ((long *)p)[i] = (long)v; break;
    case MPI_CHAR: ((char *)p)[i] = (char)v; break;
    default: ((double *)p)[i] = v; break;
    }
}
int MPI_Allreduce(const void *s, void *r, int count, MPI_Datatype t, MPI_Op op, MPI_Comm comm) {
    if (comm == MPI_COMM_SELF || job_size() == 1) {
        if (s != r) memcpy(r, s, (size_t)count * (size_t)mpi_esize(t));
        return 0;
    }
    if (comm_ready()) return 1;
    const int how = op == MPI_SUM ? 0 : (op == MPI_MAX ? 1 : 2);
    for (int i0 = 0; i0 < count; i0 += 1024) {  // sums in rank order on every rank: the same bits everywhere
        const int c = std::min(1024, count - i0);
        double v[1024];
        for (int i = 0; i < c; i++) v[i] = mpi_load(s, i0 + i, t);
        slab_detail::host_reduce(&sc, v, c, how);
        for (int i = 0; i < c; i++) mpi_store(r, i0 + i, t, v[i]);
    }
    return 0;
}
int MPI_Allgather(const void *s, int scount, MPI_Datatype st, void *r, int, MPI_Datatype, MPI_Comm comm) {
    if (comm == MPI_COMM_SELF || job_size() == 1) {
        if (s != r) memcpy(r, s, (size_t)scount * (size_t)mpi_esize(st));
        return 0;
    }
    if (comm_ready() || scount > sc.hooks.cap) return 1;
    for (int i = 0; i < scount; i++) sc.mailbox(sc.rank, 0)[i] = mpi_load(s, i, st);
    sc.barrier();
    for (int q = 0; q < sc.nranks; q++)
        for (int i = 0; i < scount; i++) mpi_store(r, q * scount + i, st, sc.mailbox(q, 0)[i]);
    sc.barrier();
    return 0;
}
int MPI_Init(int *, char ***) { return 0; }
int MPI_Finalize(void) { return 0; }
int MPI_Abort(MPI_Comm, int code) {
    fprintf(stderr, "[mpi-compat] MPI_Abort(%d)\n", code);
    exit(code ? code : 1);
}
int MPI_Type_size(MPI_Datatype t, int *size) {
    *size = mpi_esize(t);
    return 0;
}
int MPI_Type_vector(int count, int blocklength, int stride, MPI_Datatype oldtype, MPI_Datatype *newtype) {
    vec_types().push_back({count, blocklength, stride, mpi_esize(oldtype)});
    *newtype = 1000 + (int)vec_types().size() - 1;
    return 0;
}
int MPI_Type_commit(MPI_Datatype *) { return 0; }
int MPI_Type_free(MPI_Datatype *t) {
    *t = 0;
    return 0;
}
int MPI_File_open(MPI_Comm, const char *filename, int amode, MPI_Info, MPI_File *fh) {
    // MPI-IO never truncates: several open/close rounds -- and several ranks, each writing through its own view --
    // build one file
    const int fd = open(filename, O_RDWR | ((amode & MPI_MODE_CREATE) ? O_CREAT : 0), 0644);
    FILE *fp = fd >= 0 ? fdopen(fd, "r+b") : nullptr;
    if (!fp) return 1;
    *fh = new _mpi_compat_file{fp, 0, 0, 0, 0, 0};
    return 0;
}
int MPI_File_close(MPI_File *fh) {
    if (fh && *fh) {
        fclose((*fh)->fp);
        delete *fh;
        *fh = nullptr;
    }
    return 0;
}
int MPI_File_delete(const char *filename, MPI_Info) { return remove(filename) ? 1 : 0; }
int MPI_File_set_view(MPI_File fh, MPI_Offset disp, MPI_Datatype, MPI_Datatype filetype, const char *, MPI_Info) {
    fh->disp = disp;
    fh->pos = 0;
    fh->vec_block = fh->vec_stride = fh->vec_esize = 0;
    if (filetype >= 1000) {
        const VecType &v = vec_types()[(size_t)(filetype - 1000)];
        fh->vec_block = v.block;
        fh->vec_stride = v.stride;
        fh->vec_esize = v.esize;
    }
    return 0;
}
int MPI_File_write(MPI_File fh, const void *buf, int count, MPI_Datatype t, MPI_Status *) {
    const int es = mpi_esize(t);
    const char *p = (const char *)buf;
    if (!fh->vec_block) {
        fseek(fh->fp, (long)(fh->disp + fh->pos * es), SEEK_SET);
        fwrite(p, (size_t)es, (size_t)count, fh->fp);
        fh->pos += count;
        return 0;
    }
    for (int done = 0; done < count;) {  // blocks of vec_block elements every vec_stride elements
        const long long blk = fh->pos / fh->vec_block, within = fh->pos % fh->vec_block;
        const int n = (int)std::min<long long>(fh->vec_block - within, count - done);
        fseek(fh->fp, (long)(fh->disp + (blk * fh->vec_stride + within) * es), SEEK_SET);
        fwrite(p + (size_t)done * es, (size_t)es, (size_t)n, fh->fp);
        done += n;
        fh->pos += n;
    }
    return 0;
}
int MPI_File_write_all(MPI_File fh, const void *buf, int count, MPI_Datatype t, MPI_Status *st) { return MPI_File_write(fh, buf, count, t, st); }
int MPI_Comm_rank(MPI_Comm comm, int *rank) {
    *rank = comm == MPI_COMM_SELF ? 0 : job_rank();
    return 0;
}
int MPI_Comm_size(MPI_Comm comm, int *size) {
    *size = comm == MPI_COMM_SELF ? 1 : job_size();
    return 0;
}
int MPI_Barrier(MPI_Comm comm) {
    if (comm == MPI_COMM_SELF || job_size() == 1) return 0;
    if (comm_ready()) return 1;
    sc.barrier();
    return 0;
}
double MPI_Wtime(void) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
PetscErrorCode PetscViewerBinaryOpen(MPI_Comm comm, const char name[], PetscFileMode mode, PetscViewer *v) {
    FILE *fp = nullptr;
    long long pos0 = 0;
    if (job_size() == 1 || comm == MPI_COMM_SELF) {
        fp = fopen(name, mode == FILE_MODE_READ ? "rb" : (mode == FILE_MODE_APPEND ? "ab" : "wb"));
    } else {  // rank 0 creates / truncates, then every rank has the file open for positioned writes of its own part
        if (mode != FILE_MODE_READ && job_rank() == 0) {
            FILE *t = fopen(name, mode == FILE_MODE_APPEND ? "ab" : "wb");
            if (t) fclose(t);
        }
        MPI_Barrier(comm);
        fp = fopen(name, mode == FILE_MODE_READ ? "rb" : "r+b");
        if (fp && mode == FILE_MODE_APPEND) {
            fseek(fp, 0, SEEK_END);
            pos0 = ftell(fp);
        }
    }
    if (!fp) return PETSC_ERR_FILE_OPEN;
    PetscViewer w = new _p_PetscViewer();
    hdr_init(w->h, CLS_VIEWER, PETSCVIEWERBINARY);
    w->fp = fp;
    w->mode = mode;
    w->ascii = false;
    w->pos = pos0;
    *v = w;
    return 0;
}
PetscErrorCode PetscViewerCreate(MPI_Comm, PetscViewer *v) {
    PetscViewer w = new _p_PetscViewer();
    hdr_init(w->h, CLS_VIEWER, PETSCVIEWERASCII);
    w->fp = nullptr;
    w->mode = FILE_MODE_WRITE;
    w->ascii = true;
    w->pos = 0;
    *v = w;
    return 0;
}
PetscErrorCode PetscViewerSetType(PetscViewer v, PetscViewerType type) {
    v->ascii = strcmp(type, PETSCVIEWERASCII) == 0;
    return 0;
}
PetscErrorCode PetscViewerFileSetMode(PetscViewer v, PetscFileMode mode) {
    v->mode = mode;
    return 0;
}
PetscErrorCode PetscViewerFileSetName(PetscViewer v, const char name[]) {
    if (v->fp) fclose(v->fp);
    if (v->ascii && v->mode != FILE_MODE_READ && job_rank() != 0) {  // an ASCII viewer prints from the first rank only
        v->fp = nullptr;
        return 0;
    }
    v->fp = fopen(name, v->mode == FILE_MODE_READ ? "r" : (v->mode == FILE_MODE_APPEND ? "a" : "w"));
    return v->fp ? 0 : PETSC_ERR_FILE_OPEN;
}
PetscErrorCode PetscViewerASCIIPrintf(PetscViewer v, const char format[], ...) {
    if (!v->fp && v->ascii && job_rank() != 0) return 0;
    if (!v->fp) return PETSC_ERR_ORDER;
    va_list ap;
    va_start(ap, format);
    vfprintf(v->fp, format, ap);
    va_end(ap);
    return 0;
}
PetscErrorCode PetscViewerDestroy(PetscViewer *v) {
    if (v && *v) {
        if ((*v)->fp) fclose((*v)->fp);
        delete *v;
        *v = nullptr;
    }
    return 0;
}
PetscErrorCode PetscRandomCreate(MPI_Comm, PetscRandom *r) {
    *r = new _p_PetscRandom();
    hdr_init((*r)->h, CLS_RANDOM, PETSCRAND48);
    (*r)->state = 0x1234ABCD330EULL;  // rand48 default seed
    return 0;
}
PetscErrorCode PetscRandomSetType(PetscRandom, PetscRandomType) { return 0; }
PetscErrorCode PetscRandomDestroy(PetscRandom *r) {
    if (r && *r) {
        delete *r;
        *r = nullptr;
    }
    return 0;
}

}  // extern "C"

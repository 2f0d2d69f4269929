// mpi_probe -- the MPI subset of include/petsc_compat/mpi.h across the ranks of a slab job, without a GPU: reductions of
// every element type the reference reduces, Allgather, and the MPI-IO pattern of MPIIO.cc (a header written by rank 0
// through MPI_COMM_SELF, then every rank's block through its own view of the shared file: contiguous floats, and a
// strided vector filetype that interleaves several fields).  tests/test_cpp_host.py checks the file.
//   slabrun -n R mpi_probe out.bin
#include <petsc.h>

#include <cstdio>
#include <cstring>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    PetscInitialize(&argc, &argv, NULL, NULL);
    int rank = 0, size = 1, bad = 0;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &size);
    // ---- reductions
    double d[3] = {0.5 * (rank + 1), -1.0 * rank, 1e-3 * rank}, ds[3];
    MPI_Allreduce(d, ds, 3, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int q = 0; q < size; q++) e0 += 0.5 * (q + 1), e1 += -1.0 * q, e2 += 1e-3 * q;
    if (ds[0] != e0 || ds[1] != e1 || ds[2] != e2) bad++;
    int im = 3 * rank + 1, imx = 0;
    MPI_Allreduce(&im, &imx, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);
    if (imx != 3 * (size - 1) + 1) bad++;
    unsigned long ul = 1000000007ul * (unsigned long)(rank + 1), uls = 0, ule = 0;
    MPI_Allreduce(&ul, &uls, 1, MPI_UNSIGNED_LONG, MPI_SUM, MPI_COMM_WORLD);
    for (int q = 0; q < size; q++) ule += 1000000007ul * (unsigned long)(q + 1);
    if (uls != ule) bad++;
    double mn = 10.0 - rank;
    MPI_Allreduce(&mn, &mn, 1, MPI_DOUBLE, MPI_MIN, MPI_COMM_WORLD);  // in place
    if (mn != 10.0 - (size - 1)) bad++;
    // ---- gather of the block sizes (MPIIO.cc:274-281)
    const unsigned long nloc = 5 + (unsigned long)rank;  // floats this rank owns
    std::vector<unsigned long> all((size_t)size);
    MPI_Allgather(&nloc, 1, MPI_UNSIGNED_LONG, all.data(), 1, MPI_UNSIGNED_LONG, MPI_COMM_WORLD);
    unsigned long before = 0, total = 0;
    for (int q = 0; q < size; q++) {
        if (all[(size_t)q] != 5 + (unsigned long)q) bad++;
        if (q < rank) before += all[(size_t)q];
        total += all[(size_t)q];
    }
    // ---- the file: header by rank 0, then one contiguous dataset and one of 3 interleaved fields
    const char *fn = argv[1];
    const char hdr[] = "probe v1\n";
    MPI_File fh;
    if (rank == 0) {
        MPI_File_delete(fn, MPI_INFO_NULL);
        if (MPI_File_open(MPI_COMM_SELF, fn, MPI_MODE_CREATE | MPI_MODE_WRONLY, MPI_INFO_NULL, &fh)) bad++;
        MPI_File_set_view(fh, 0, MPI_CHAR, MPI_CHAR, (char *)"native", MPI_INFO_NULL);
        MPI_File_write(fh, hdr, (int)strlen(hdr), MPI_CHAR, MPI_STATUS_IGNORE);
        MPI_File_close(&fh);
    }
    MPI_Barrier(MPI_COMM_WORLD);
    MPI_Offset off = (MPI_Offset)strlen(hdr);
    std::vector<float> blk(nloc);
    for (unsigned long i = 0; i < nloc; i++) blk[i] = (float)(100 * rank + (int)i);
    if (MPI_File_open(MPI_COMM_WORLD, fn, MPI_MODE_CREATE | MPI_MODE_WRONLY, MPI_INFO_NULL, &fh)) bad++;
    MPI_File_set_view(fh, off + (MPI_Offset)(4 * before), MPI_FLOAT, MPI_FLOAT, (char *)"native", MPI_INFO_NULL);
    MPI_File_write_all(fh, blk.data(), (int)nloc, MPI_FLOAT, MPI_STATUS_IGNORE);
    MPI_File_close(&fh);
    off += (MPI_Offset)(4 * total);
    // 3 fields, field-major in the file ([field][all ranks' blocks]), rank-major in memory (MPIIO.cc:587-640)
    const int nf = 3;
    std::vector<float> fields((size_t)nf * nloc);
    for (int f = 0; f < nf; f++)
        for (unsigned long i = 0; i < nloc; i++) fields[(size_t)f * nloc + i] = (float)(1000 * (f + 1) + 100 * rank + (int)i);
    MPI_Datatype ft;
    MPI_Type_vector(nf, (int)nloc, (int)total, MPI_FLOAT, &ft);
    MPI_Type_commit(&ft);
    if (MPI_File_open(MPI_COMM_WORLD, fn, MPI_MODE_CREATE | MPI_MODE_WRONLY, MPI_INFO_NULL, &fh)) bad++;
    MPI_File_set_view(fh, off + (MPI_Offset)(4 * before), MPI_FLOAT, ft, (char *)"native", MPI_INFO_NULL);
    MPI_File_write_all(fh, fields.data(), nf * (int)nloc, MPI_FLOAT, MPI_STATUS_IGNORE);
    MPI_File_close(&fh);
    MPI_Type_free(&ft);
    MPI_Barrier(MPI_COMM_WORLD);
    printf("rank %d of %d %s\n", rank, size, bad ? "FAILED" : "OK");
    PetscFinalize();
    return bad ? 1 : 0;
}

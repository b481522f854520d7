/*
 * csr.h -- Compressed Sparse Row matrices: container, Matrix Market loader,
 * synthetic generator, CPU benchmarks and the GPU (HIP) benchmark wrappers.
 *
 * Host API kept from the reference (include/csr.h:7-49): same struct layout
 * and field names, same function names for loader / free / CPU benches, and
 * the same `int f(const sparse_csr*, const double *x, bench_xxx *out)` shape
 * for every benchmark.  The five GPU wrappers are the MI355X counterparts of
 * the reference's bench_csr_cuda_* (csr.h:40-49).
 */
#ifndef SPMV_CSR_H
#define SPMV_CSR_H

#include <stdint.h>

#include "utils.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sparse_matrix_csr {
    char name[MAX_NAME]; /* file base name without ".mtx" */
    int M, N, NZ;        /* rows, columns, stored entries */
    int *IRP;            /* [M+1] row start offsets, IRP[0] = 0 */
    int *JA;             /* [NZ] 0-based columns, file order inside a row */
    double *AS;          /* [NZ] values */
} sparse_csr;

static inline void init_csr(sparse_csr *A, const char *name, int M, int N,
                            int NZ, int *IRP, int *JA, double *AS) {
    snprintf(A->name, sizeof A->name, "%s", name);
    A->M = M;
    A->N = N;
    A->NZ = NZ;
    A->IRP = IRP;
    A->JA = JA;
    A->AS = AS;
}

/*
 * Matrix Market -> CSR, semantics of the reference loader (csr.c:31-171):
 *  - accepts "matrix coordinate real|pattern general|symmetric|...";
 *    complex / integer / array -> ERR_PTR(-EINVAL);
 *  - "symmetric" mirrors every off-diagonal entry; skew-symmetric and
 *    hermitian are read as general;
 *  - pattern entries get the value 1.0;
 *  - entries keep FILE order inside each row (a mirrored entry is placed
 *    right after the entry that produced it), duplicates are kept;
 *  - index out of range -> -ERANGE; truncated / unparsable data -> -EIO;
 *    unreadable file -> -errno of fopen;
 *  - more than INT_MAX stored entries -> -EOVERFLOW (the reference
 *    silently truncates, csr.c:153).
 * Single pass over an in-memory image of the file (the reference parses the
 * text twice with fscanf); files of >= 100k entries whose tokens are all
 * plain numbers are tokenised and converted by all OpenMP threads.
 * Returns ERR_PTR(code) on failure, never NULL.
 */
sparse_csr *io_load_csr(const char *path);

/*
 * Binary sidecar of a loaded matrix (new): header + the three
 * CSR arrays.  io_load_csr_cached(path) reads "<path>.bin" only when its
 * header names exactly this text file (size and mtime to the nanosecond),
 * else parses the text and replaces the sidecar (written under a temporary
 * name and rename()d into place, so concurrent loaders never see a torn
 * file).  Every load validates the arrays (IRP monotone from 0 to NZ, every
 * JA in [0, N)): csr_load_bin returns ERR_PTR(-EINVAL) for a foreign header,
 * -EIO for a truncated file, -EILSEQ for arrays that are not a CSR matrix,
 * -ESTALE (cached path only, then re-parsed) for another file's sidecar.
 *
 * Contract for writers of .mtx files: write under a temporary name and
 * rename() into place.  io_load_csr maps the text (MAP_PRIVATE); a file that
 * is truncated or rewritten in place while a loader parses it would end that
 * loader with SIGBUS.  (A file modified within the last two seconds is read
 * into memory instead of mapped, as a second line of defence.)
 */
int csr_save_bin(const sparse_csr *A, const char *path);
sparse_csr *csr_load_bin(const char *path);
sparse_csr *io_load_csr_cached(const char *path);

/* basename without a trailing ".mtx", at most MAX_NAME-1 chars. */
void extract_matrix_name(const char *path, char *name_out);

/* Allocate an empty CSR with 64 B aligned arrays (IRP zeroed). */
sparse_csr *csr_alloc(const char *name, int M, int N, int NZ);

/* Synthetic matrix (include/spmv_synth.h families), built in parallel.
 * row0 = global index of the first local row (multi-GPU shards). */
sparse_csr *csr_generate(int kind, int M, int N, int K, int64_t W,
                         int64_t row0, uint64_t seed);

/* Result check of a full-size synthetic run without materialising the
 * matrix: for each of the n GLOBAL rows in `rows`, dot[k] = sum_j a_ij *
 * synth_x(xseed, col_j) accumulated left to right like the serial CSR kernel
 * (reference csr.c:201-216) and scale[k] = sum_j |a_ij x_j|.  One serial
 * call, no OpenMP team (n is a few hundred).  Returns 0 or -EINVAL. */
int csr_synth_row_dots(int kind, int N, int K, int64_t W, uint64_t seed,
                       uint64_t xseed, const int64_t *rows, int n,
                       double *dot, double *scale);

/* Rows [r0, r1) of A as an independent matrix with GLOBAL columns. */
sparse_csr *csr_row_slice(const sparse_csr *A, int r0, int r1);

void csr_free(sparse_csr *A);

/*
 * Row-range partitions.
 * partition_rows_nnz: the reference's greedy nnz-balanced cut
 *   (csr.c:218-276); *parts may shrink; returns malloc'ed starts[*parts+1].
 * partition_rows_even: equal row counts rounded up to `align` rows
 *   (align = 32 keeps HLL hack blocks whole); starts[parts+1].
 */
int *partition_rows_nnz(const sparse_csr *A, int *parts);
int *partition_rows_even(int M, int parts, int align);
/*
 * The multi-GPU form of the nnz-balanced cut (modelled on the
 * reference's partition_csr_rows, csr.c:218-276): `parts` contiguous row
 * ranges of near-equal entry counts whose boundaries are multiples of
 * `align` rows (32: no HLL hack block straddles two GPUs).  Cut k is the
 * aligned boundary whose prefix entry count is nearest to k/parts of the
 * total -- prefix targets instead of the reference's per-range running
 * count, which hands every overshoot to the last range.  `parts` never
 * shrinks; no range is empty while M >= parts * align (with fewer aligned
 * blocks than parts the trailing ranges are empty, as in
 * partition_rows_even).  IRP = row offsets of the whole matrix (M + 1).
 * Returns malloc'ed starts[parts+1] or ERR_PTR(-EINVAL / -ENOMEM).
 */
int *partition_rows_nnz_aligned(const int *IRP, int M, int parts, int align);
/* the same cut for a synthetic family (spmv_synth.h) of M global rows,
 * from the generator's row lengths -- no matrix is materialised */
int *partition_synth_rows_nnz(int kind, int M, int N, int K, int64_t W,
                              uint64_t seed, int parts, int align);

/* ---- CPU benchmarks (reference csr.c:342-380) ---- */
int bench_csr_serial(const sparse_csr *A, const double *x, bench *out);
int bench_csr_omp_guided(const sparse_csr *A, const double *x, bench_omp *out);
int bench_csr_omp_nnz_balancing(const sparse_csr *A, const double *x,
                                bench_omp *out);

/* ---- MI355X benchmarks (one per HIP kernel, hip_csr.h) ----
 * Return 0, -ENOMEM, or a negative errno mapped from a HIP failure
 * (-ENODEV when no GPU is present: there is NO CPU fallback). */
int bench_csr_hip_thread_row(const sparse_csr *A, const double *x,
                             bench_hip *out);
int bench_csr_hip_wave_row(const sparse_csr *A, const double *x,
                           bench_hip *out);
int bench_csr_hip_subwave_row(const sparse_csr *A, const double *x,
                              bench_hip *out);
int bench_csr_hip_block_row(const sparse_csr *A, const double *x,
                            bench_hip *out);
int bench_csr_hip_stream(const sparse_csr *A, const double *x, bench_hip *out);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_CSR_H */

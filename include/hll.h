/*
 * hll.h -- HLL ("hacked ELLPACK"): the matrix is cut into hack blocks of
 * HACK_SIZE consecutive rows; each block is a small ELLPACK matrix padded
 * to the longest row OF THAT BLOCK.
 *
 * Host API kept from the reference (include/hll.h:10-70): struct layouts,
 * HACK_SIZE, csr_to_hll / hll_free, CPU benches, and one GPU bench wrapper
 * per kernel.
 *
 * Block storage (reference hll.c:73-90):
 *   pad slots    JA = -1, AS = 0.0
 *   row-major    slot(i, j) = i * max_NZ + j
 *   col-major    slot(i, j) = j * M + i      (M = rows of THIS block; the
 *                                             tail block has M < HACK_SIZE)
 * Unlike the reference (one malloc pair per block, hll.c:60-61) the blocks
 * produced by csr_to_hll() live in two contiguous slabs -- block b starts at
 * slot offset sum_{k<b} M_k * max_NZ_k -- so the device upload is two large
 * copies.  blocks[b].JA / .AS still point at each block, as before.
 */
#ifndef SPMV_HLL_H
#define SPMV_HLL_H

#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>

#include "csr.h"
#include "utils.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HACK_SIZE 32

typedef struct {
    int M, N, NZ; /* rows of the block, matrix columns, true entries */
    int max_NZ;   /* padded row length */
    int *JA;      /* [M * max_NZ] */
    double *AS;   /* [M * max_NZ] */
} ellpack_block;

typedef struct {
    char name[MAX_NAME];
    int M, N, NZ; /* NZ = true entries (GFLOP/s use this, hll.c:121) */
    int hack_size;
    int num_blocks;
    ellpack_block *blocks;
} sparse_hll;

static inline void init_ellpack_block(ellpack_block *b, int M, int N, int NZ,
                                      int max_NZ) {
    b->M = M;
    b->N = N;
    b->NZ = NZ;
    b->max_NZ = max_NZ;
    b->JA = NULL;
    b->AS = NULL;
}

static inline void init_hll(sparse_hll *H, const char *name, int M, int N,
                            int NZ, int num_blocks) {
    snprintf(H->name, sizeof H->name, "%s", name);
    H->M = M;
    H->N = N;
    H->NZ = NZ;
    H->hack_size = HACK_SIZE;
    H->num_blocks = num_blocks;
    H->blocks = NULL;
}

/* CSR -> HLL.  Returns ERR_PTR(-ENOMEM) on failure (never NULL; the
 * reference documents NULL but returns ERR_PTR, hll.h:52 vs hll.c:26). */
sparse_hll *csr_to_hll(const sparse_csr *A, bool is_col_major);

/* Release an HLL made by csr_to_hll().  Also accepts a matrix whose blocks
 * were allocated one by one (reference style). */
void hll_free(sparse_hll *H);

/* Total stored slots, padding included: sum_b M_b * max_NZ_b. */
int64_t hll_num_slots(const sparse_hll *H);

/* 1 if block storage is one contiguous slab pair in block order. */
int hll_is_contiguous(const sparse_hll *H);
/* copy the blocks of a block-by-block matrix into two slabs (block b at
 * off[b], off[nb] = hll_num_slots(H)), in parallel */
void hll_pack_slabs(const sparse_hll *H, const int64_t *off, int *ja,
                    double *as);

/* ---- CPU benchmarks (reference hll.c:214-224); row-major input ---- */
int bench_hll_serial(const sparse_hll *H, const double *x, bench *out);
int bench_hll_omp(const sparse_hll *H, const double *x, bench_omp *out);
/* col-major counterpart of bench_hll_serial (reference keeps it unused,
 * hll.c:152-176) */
int bench_hll_serial_col_major(const sparse_hll *H, const double *x,
                               bench *out);

/* ---- MI355X benchmarks (one per HIP kernel, hip_hll.h) ----
 * *_row_major / subwave_row take a row-major H, the other two col-major,
 * exactly as the reference pairs them (main.c:324-325). */
int bench_hll_hip_threads_row_major(const sparse_hll *H, const double *x,
                                    bench_hip *out);
int bench_hll_hip_threads_col_major(const sparse_hll *H, const double *x,
                                    bench_hip *out);
int bench_hll_hip_wave_block(const sparse_hll *H, const double *x,
                             bench_hip *out);
int bench_hll_hip_subwave_row(const sparse_hll *H, const double *x,
                              bench_hip *out);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_HLL_H */

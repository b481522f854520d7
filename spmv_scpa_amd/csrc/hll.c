/*
 * hll.c -- CSR -> HLL conversion, CPU HLL kernels and the GPU benchmark
 * wrappers (API: include/hll.h).
 *
 * Behavioural reference: src/hll.c of 0xmenna/spmv-scpa (cited per
 * function).  Storage differs: all blocks of a matrix share one JA slab and
 * one AS slab (block b at slot offset off[b]) so the device upload is two
 * copies instead of three per hack block (reference cuda_hll.cu:161-206);
 * each block's JA/AS pointers address its part of the slabs.
 */
#include <errno.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#include "err.h"
#include "hip_hll.h"
#include "hll.h"
#include "spmv_engine.h"

static int block_rows(int M, int b) {
    int r0 = b * HACK_SIZE;
    return (r0 + HACK_SIZE <= M ? HACK_SIZE : M - r0);
}

sparse_hll *csr_to_hll(const sparse_csr *A, bool is_col_major) {
    if (IS_ERR_OR_NULL(A))
        return ERR_PTR(-EINVAL);
    const int M = A->M, nb = (M + HACK_SIZE - 1) / HACK_SIZE;

    sparse_hll *H = malloc(sizeof *H);
    int *ja_slab = NULL;
    double *as_slab = NULL;
    if (!H)
        return ERR_PTR(-ENOMEM);
    init_hll(H, A->name, M, A->N, A->NZ, nb);
    H->blocks = aligned_malloc(((size_t)nb + 1) * sizeof(ellpack_block));
    int64_t *off = malloc(((size_t)nb + 1) * sizeof *off);
    if (!H->blocks || !off)
        goto nomem;

    /* pass 1 (parallel): per-block longest row and entry count */
#pragma omp parallel for schedule(static) num_threads(spmv_host_threads())
    for (int b = 0; b < nb; ++b) {
        int r0 = b * HACK_SIZE, rows = block_rows(M, b);
        int longest = 0;
        for (int i = 0; i < rows; ++i) {
            int len = A->IRP[r0 + i + 1] - A->IRP[r0 + i];
            if (len > longest)
                longest = len;
        }
        init_ellpack_block(&H->blocks[b], rows, A->N,
                           A->IRP[r0 + rows] - A->IRP[r0], longest);
    }
    off[0] = 0;
    for (int b = 0; b < nb; ++b)
        off[b + 1] = off[b] + (int64_t)H->blocks[b].M * H->blocks[b].max_NZ;
    if (nb > 0) { /* with no block nothing could point at (and free) them */
        ja_slab = aligned_malloc((size_t)off[nb] * sizeof(int));
        as_slab = aligned_malloc((size_t)off[nb] * sizeof(double));
        if (!ja_slab || !as_slab)
            goto nomem;
    }

    /* pass 2 (parallel): pad with (-1, 0.0), then place the entries
     * (reference hll.c:73-90) */
#pragma omp parallel for schedule(dynamic, 256) num_threads(spmv_host_threads())
    for (int b = 0; b < nb; ++b) {
        ellpack_block *blk = &H->blocks[b];
        int r0 = b * HACK_SIZE, rows = blk->M, width = blk->max_NZ;
        int *ja = ja_slab + off[b];
        double *as = as_slab + off[b];
        blk->JA = ja;
        blk->AS = as;
        for (int i = 0; i < rows; ++i) {
            const int st = A->IRP[r0 + i];
            const int len = A->IRP[r0 + i + 1] - st;
            if (is_col_major) {
                for (int j = 0; j < len; ++j) {
                    ja[(size_t)j * rows + i] = A->JA[st + j];
                    as[(size_t)j * rows + i] = A->AS[st + j];
                }
                for (int j = len; j < width; ++j) {
                    ja[(size_t)j * rows + i] = -1;
                    as[(size_t)j * rows + i] = 0.0;
                }
            } else {
                int *rj = ja + (size_t)i * width;
                double *ra = as + (size_t)i * width;
                memcpy(rj, A->JA + st, (size_t)len * sizeof(int));
                memcpy(ra, A->AS + st, (size_t)len * sizeof(double));
                for (int j = len; j < width; ++j) {
                    rj[j] = -1;
                    ra[j] = 0.0;
                }
            }
        }
    }
    free(off);
    return H;

nomem:
    free(off);
    free(ja_slab);
    free(as_slab);
    free(H->blocks);
    free(H);
    return ERR_PTR(-ENOMEM);
}

/*
 * A matrix from csr_to_hll() is recognised by its contiguous block layout:
 * block 0 then owns both slabs.  A matrix assembled block by block
 * (reference style, one allocation pair per block) is freed block by block.
 */
void hll_free(sparse_hll *H) {
    if (IS_ERR_OR_NULL(H))
        return;
    if (H->num_blocks > 0 && hll_is_contiguous(H)) {
        free(H->blocks[0].JA);
        free(H->blocks[0].AS);
    } else {
        for (int b = 0; b < H->num_blocks; ++b) {
            free(H->blocks[b].JA);
            free(H->blocks[b].AS);
        }
    }
    free(H->blocks);
    free(H);
}

int64_t hll_num_slots(const sparse_hll *H) {
    int64_t s = 0;
    for (int b = 0; b < H->num_blocks; ++b)
        s += (int64_t)H->blocks[b].M * H->blocks[b].max_NZ;
    return s;
}

int hll_is_contiguous(const sparse_hll *H) {
    if (H->num_blocks == 0)
        return 1;
    const int *ja = H->blocks[0].JA;
    const double *as = H->blocks[0].AS;
    int64_t at = 0;
    for (int b = 0; b < H->num_blocks; ++b) {
        if (H->blocks[b].JA != ja + at || H->blocks[b].AS != as + at)
            return 0;
        at += (int64_t)H->blocks[b].M * H->blocks[b].max_NZ;
    }
    return 1;
}

/*
 * The blocks of a matrix assembled block by block (what the reference's
 * csr_to_hll returns: one allocation pair per hack block, hll.c:56-70) copied
 * into two slabs in block order, in parallel: the device upload of such a
 * matrix is then two large copies instead of two per block (312 500 blocks at
 * config 3: 625 000 hipMemcpy calls, seconds per one-shot call; the reference
 * pays 3 per block, cuda_hll.cu:161-206).  ja / as: caller's buffers of
 * hll_num_slots(H) elements.
 */
void hll_pack_slabs(const sparse_hll *H, const int64_t *off, int *ja,
                    double *as) {
    /* a copy loop: a handful of threads saturate the memory system, and a
     * team sized from the affinity mask (256 on the GPU box) under a 16-CPU
     * cgroup quota spends its time being throttled */
    int team = spmv_host_threads();
    if (team > 8)
        team = 8;
#pragma omp parallel for schedule(dynamic, 512) num_threads(team)
    for (int b = 0; b < H->num_blocks; ++b) {
        const size_t n = (size_t)(off[b + 1] - off[b]);
        if (!n)
            continue;
        memcpy(ja + off[b], H->blocks[b].JA, n * sizeof(int));
        memcpy(as + off[b], H->blocks[b].AS, n * sizeof(double));
    }
}

/* ------------------------------------------------------------------ */
/* CPU kernels (reference hll.c:127-211): pads are skipped              */
/* ------------------------------------------------------------------ */

typedef double (*hll_kernel_fn)(const sparse_hll *, const double *, double *,
                                void *);

static inline void block_rows_major(const ellpack_block *blk, const double *x,
                                    double *y) {
    for (int i = 0; i < blk->M; ++i) {
        const int *rj = blk->JA + (size_t)i * blk->max_NZ;
        const double *ra = blk->AS + (size_t)i * blk->max_NZ;
        double acc = 0.0;
        for (int j = 0; j < blk->max_NZ; ++j)
            if (rj[j] != -1)
                acc += ra[j] * x[rj[j]];
        y[i] = acc;
    }
}

static double hll_spmv_serial(const sparse_hll *H, const double *x, double *y,
                              void *arg) {
    (void)arg;
    double t0 = now();
    for (int b = 0; b < H->num_blocks; ++b)
        block_rows_major(&H->blocks[b], x, y + (size_t)b * HACK_SIZE);
    return now() - t0;
}

static double hll_spmv_serial_cm(const sparse_hll *H, const double *x,
                                 double *y, void *arg) {
    (void)arg;
    double t0 = now();
    for (int b = 0; b < H->num_blocks; ++b) {
        const ellpack_block *blk = &H->blocks[b];
        for (int i = 0; i < blk->M; ++i) {
            double acc = 0.0;
            for (int j = 0; j < blk->max_NZ; ++j) {
                size_t t = (size_t)j * blk->M + i;
                if (blk->JA[t] != -1)
                    acc += blk->AS[t] * x[blk->JA[t]];
            }
            y[(size_t)b * HACK_SIZE + i] = acc;
        }
    }
    return now() - t0;
}

static double hll_spmv_omp(const sparse_hll *H, const double *x, double *y,
                           void *arg) {
    int threads = *(const int *)arg;
    double t0 = omp_get_wtime();
#pragma omp parallel for schedule(guided) num_threads(threads)
    for (int b = 0; b < H->num_blocks; ++b)
        block_rows_major(&H->blocks[b], x, y + (size_t)b * HACK_SIZE);
    return (omp_get_wtime() - t0) * 1e3;
}

static int run_hll_bench(const sparse_hll *H, const double *x, bench *out,
                         void *arg, hll_kernel_fn fn) {
    vec y = vec_create((size_t)H->M);
    if (!y.data)
        return -ENOMEM;
    double ms = fn(H, x, y.data, arg);
    if (ms < 0.0) {
        vec_put(&y);
        return (int)ms;
    }
    out->duration_ms = ms;
    out->gflops = compute_gflops(ms, H->NZ); /* true entries, hll.c:121 */
    out->data = y;
    return 0;
}

int bench_hll_serial(const sparse_hll *H, const double *x, bench *out) {
    return run_hll_bench(H, x, out, NULL, hll_spmv_serial);
}

int bench_hll_serial_col_major(const sparse_hll *H, const double *x,
                               bench *out) {
    return run_hll_bench(H, x, out, NULL, hll_spmv_serial_cm);
}

int bench_hll_omp(const sparse_hll *H, const double *x, bench_omp *out) {
    snprintf(out->name, sizeof out->name, "omp_guided");
    if (out->num_threads < 1)
        return -EINVAL;
    return run_hll_bench(H, x, &out->bench, &out->num_threads, hll_spmv_omp);
}

/* ------------------------------------------------------------------ */
/* GPU wrappers (counterparts of reference hll.c:226-256)               */
/* ------------------------------------------------------------------ */

static int run_hll_hip(const sparse_hll *H, const double *x, bench_hip *out,
                       hll_kernel_fn fn) {
    spmv_launch_opts opts;
    memset(&opts, 0, sizeof opts);
    opts.waves_per_block = out->waves_per_block;
    return run_hll_bench(H, x, &out->bench, &opts, fn);
}

int bench_hll_hip_threads_row_major(const sparse_hll *H, const double *x,
                                    bench_hip *out) {
    return run_hll_hip(H, x, out, hll_spmv_hip_threads_row_major);
}

int bench_hll_hip_threads_col_major(const sparse_hll *H, const double *x,
                                    bench_hip *out) {
    return run_hll_hip(H, x, out, hll_spmv_hip_threads_col_major);
}

int bench_hll_hip_wave_block(const sparse_hll *H, const double *x,
                             bench_hip *out) {
    return run_hll_hip(H, x, out, hll_spmv_hip_wave_block);
}

int bench_hll_hip_subwave_row(const sparse_hll *H, const double *x,
                              bench_hip *out) {
    return run_hll_hip(H, x, out, hll_spmv_hip_subwave_row);
}

/*
 * hip_hll.h -- C-ABI of the HLL SpMV kernels for MI355X (gfx950).
 *
 * Replaces the reference's include/cuda_hll.h:10-22; same seam and the same
 * contract as hip_csr.h.  On upload the pad slots (JA = -1) of each row are
 * rewritten to the previous valid column of that row, or 0 for an empty
 * row, as the reference's upload does (cuda_hll.cu:173-195), so the kernels
 * run without a pad branch (the pad value is 0.0).
 *
 * Kernel ids (reference main.c:310-315):
 *   0 threads_row_major  one lane per row, row-major blocks
 *   1 threads_col_major  one lane per row, col-major blocks staged through
 *                        LDS with 16 B/lane coalesced loads (north star)
 *   2 wave_block         one wavefront per pair of hack blocks, col-major,
 *                        direct loads
 *   3 subwave_row        16 lanes per row, row-major, __shfl_down tree
 */
#ifndef SPMV_HIP_HLL_H
#define SPMV_HIP_HLL_H

#include "hll.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SPMV_NUM_HLL_KERNELS 4

/* as set_csr_waves_per_block: 1..16, 0 = back to the size-based default */
void set_hll_waves_per_block(int waves);

double hll_spmv_hip_threads_row_major(const sparse_hll *H, const double *x,
                                      double *y, void *arg);
double hll_spmv_hip_threads_col_major(const sparse_hll *H, const double *x,
                                      double *y, void *arg);
double hll_spmv_hip_wave_block(const sparse_hll *H, const double *x, double *y,
                               void *arg);
double hll_spmv_hip_subwave_row(const sparse_hll *H, const double *x,
                                double *y, void *arg);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_HIP_HLL_H */

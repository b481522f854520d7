/*
 * spmv_ref_abi.h -- the reference's own plugin symbols, exported by
 * libspmv_scpa_amd.so so that the reference's UNMODIFIED csr.c / hll.c /
 * main.c link against this library instead of its CUDA translation units.
 *
 * These are exactly the 11 symbols the reference's host layer binds
 * (reference include/cuda_csr.h:10-25 and include/cuda_hll.h:10-22, called
 * from csr.c:382-415 and hll.c:226-256).  Each one forwards to the MI355X
 * kernel that fills the same slot of the driver tables (main.c:259-263,
 * 310-315); see hip_csr.h / hip_hll.h for what runs.  `wppb` keeps the
 * reference's 2/4/8 sweep but counts 64-lane wavefronts.
 *
 * Not a CUDA shim: no CUDA API is emulated; these are names of the seam.
 */
#ifndef SPMV_REF_ABI_H
#define SPMV_REF_ABI_H

#include "csr.h"
#include "hll.h"

#ifdef __cplusplus
extern "C" {
#endif

/* the reference's name for the GPU benchmark record (utils.h:44-47) */
typedef bench_hip bench_cuda;

void set_csr_warps_per_block(int wppb);          /* cuda_csr.h:10 */
double csr_spmv_cuda_thread_row(const sparse_csr *A, const double *x,
                                double *y, void *unused);       /* :12 */
double csr_spmv_cuda_warp_row(const sparse_csr *A, const double *x, double *y,
                              void *unused);                    /* :15 */
double csr_spmv_cuda_halfwarp_row(const sparse_csr *A, const double *x,
                                  double *y, void *unused);     /* :18 */
double csr_spmv_cuda_block_row(const sparse_csr *A, const double *x, double *y,
                               void *unused);                   /* :21 */
double csr_spmv_cuda_halfwarp_row_text(const sparse_csr *A, const double *x,
                                       double *y, void *unused); /* :24 */

void set_hll_warps_per_block(int wppb);          /* cuda_hll.h:10 */
double hll_spmv_cuda_threads_row_major(const sparse_hll *H, const double *x,
                                       double *y, void *unused); /* :12 */
double hll_spmv_cuda_threads_col_major(const sparse_hll *H, const double *x,
                                       double *y, void *unused); /* :15 */
double hll_spmv_cuda_warp_block(const sparse_hll *H, const double *x,
                                double *y, void *unused);        /* :18 */
double hll_spmv_cuda_halfwarp_row(const sparse_hll *H, const double *x,
                                  double *y, void *unused);      /* :21 */

#ifdef __cplusplus
}
#endif
#endif /* SPMV_REF_ABI_H */

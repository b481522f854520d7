/*
 * hip_csr.h -- C-ABI of the CSR SpMV kernels for MI355X (gfx950).
 *
 * Replaces the reference's include/cuda_csr.h:10-25.  Same seam: plain
 * extern "C" functions of the shape
 *
 *     double f(const sparse_csr *A, const double *x, double *y, void *arg);
 *
 * that the host layer passes as a function pointer to its benchmark harness
 * (reference csr.c:182-199).  Contract:
 *   - A, x (A->N doubles) and y (A->M doubles) are HOST memory owned by the
 *     caller; the callee uploads, launches, downloads y and releases every
 *     device allocation before returning (reference cuda_csr.cu:180-205);
 *   - the return value is the kernel time in milliseconds measured with a
 *     hipEvent pair on the launch stream (reference cuda_timer.cu:15-21),
 *     or a NEGATIVE errno (-ENODEV no GPU, -ENOMEM, -EIO launch/copy fault):
 *     the reference has no error channel; here every HIP call is checked;
 *   - `arg` may be NULL, or point to a spmv_launch_opts (spmv_engine.h)
 *     overriding the per-process default set by set_csr_waves_per_block().
 *
 * Kernel ids (index into the driver table, as in reference main.c:259-263):
 *   0 thread_row    one lane per row
 *   1 wave_row      one 64-lane wavefront per row, __shfl_down tree
 *   2 subwave_row   G = 2..32 lanes per row (G from mean row length),
 *                   segmented __shfl_down reduction inside the wave
 *   3 block_row     one workgroup per row (very long rows)
 *   4 stream        nnz-balanced: a workgroup streams a fixed nnz budget
 *                   through LDS, rows are reduced from LDS (takes the slot
 *                   of the reference's texture-cache variant, which has no
 *                   CDNA counterpart)
 */
#ifndef SPMV_HIP_CSR_H
#define SPMV_HIP_CSR_H

#include "csr.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SPMV_NUM_CSR_KERNELS 5

/* Wavefronts (64 lanes) per workgroup for subsequent calls that do not name
 * one: 1..16 (larger values are clamped to 16); 0 or less returns to the
 * default, which is picked by matrix size (4 below 2M rows, 8 from there).
 * (reference: set_csr_warps_per_block) */
void set_csr_waves_per_block(int waves);

/*
 * Opt-in "keep the last upload" behind the seam -- process state, like the
 * setter above; covers the CSR and the HLL entry points (and the reference's
 * names of spmv_ref_abi.h).  Default 0: every call uploads, launches,
 * downloads and releases, the reference's contract (cuda_csr.cu:180-205).
 *   level 1  the device copy of the matrix uploaded LAST through the seam is
 *            kept (one slot each for CSR, row-major HLL, col-major HLL) and
 *            reused by a call with the same struct pointer, shape and 64-bit
 *            fingerprint (array pointers, sizes, heads, tails and 64 strided
 *            samples of IRP / JA / AS; HLL: first, middle and last hack
 *            block); a different matrix replaces its slot.  x is uploaded and
 *            y downloaded on every call.
 *   level 2  also skips the upload of x when the same x (pointer + sampled
 *            fingerprint) is already on the device.
 *   level 3  like 2, but the fingerprints are a 64-bit hash of EVERY byte of
 *            IRP / JA / AS (HLL: of every hack block) and of x, recomputed on
 *            every call: the level that cannot return a stale result
 *            whatever was edited in place, at the price of reading the
 *            matrix once per call on the host (config 2, 212 MB: tens of
 *            milliseconds single-threaded, a few with the host's cores --
 *            DESIGN.md has the measured figure; still no PCIe traffic).
 *   level 0  releases what is held (call it before the process exits when
 *            spmv_live_handles() matters to you).
 * A copy belongs to the HIP device that was current when it was uploaded: a
 * call made with another device current is a miss (and re-uploads there).
 * CONTRACT of levels 1 and 2: THE CALLER PROMISES not to modify a matrix
 * (level 1) or x (level 2) in place between calls in a way the samples miss
 * -- the reference's driver never writes to either (main.c:258-354: 27 calls
 * per matrix on one A, one x).  A caller that does edit in place either runs
 * level 3 or says so: spmv_seam_cache_invalidate(ptr) drops the copy of the
 * matrix struct `ptr` (sparse_csr* / sparse_hll*), or marks the uploaded x
 * stale when `ptr` is that x; NULL drops everything held and keeps the
 * level.  Results and the returned kernel time are those of the uncached
 * call.
 */
void spmv_seam_cache(int level);
/* matrices held now (0..3); *hits / *misses (may be NULL) since start */
int spmv_seam_cache_stats(long *hits, long *misses);
/* -> slots touched (see above) */
int spmv_seam_cache_invalidate(const void *host);

double csr_spmv_hip_thread_row(const sparse_csr *A, const double *x, double *y,
                               void *arg);
double csr_spmv_hip_wave_row(const sparse_csr *A, const double *x, double *y,
                             void *arg);
double csr_spmv_hip_subwave_row(const sparse_csr *A, const double *x,
                                double *y, void *arg);
double csr_spmv_hip_block_row(const sparse_csr *A, const double *x, double *y,
                              void *arg);
double csr_spmv_hip_stream(const sparse_csr *A, const double *x, double *y,
                           void *arg);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_HIP_CSR_H */

/*
 * spmv_engine.h -- persistent device-side API under the one-shot entry
 * points of hip_csr.h / hip_hll.h.
 *
 * The reference uploads the whole matrix on every call and times a single
 * un-warmed launch (cuda_csr.cu:180-234; one cudaMalloc pair and three
 * copies PER HACK BLOCK for HLL, cuda_hll.cu:161-206).  Here a matrix is
 * uploaded once into a device handle, launched any number of times on a
 * caller-supplied HIP stream with device-resident x and y, and released.
 * The one-shot functions are thin wrappers over this API.
 *
 * Plain C ABI: opaque handles, raw device pointers (e.g. a torch tensor's
 * data_ptr()), `void *stream` = hipStream_t (NULL = default stream).
 * Every function returns 0 or a negative errno; nothing falls back to the
 * CPU: without a GPU they return -ENODEV.
 *
 * Device layout
 *   CSR   IRP int32[M+1], JA int32[NZ], AS f64[NZ]          (as on host)
 *         + row-block table for the nnz-balanced stream kernel
 *   HLL   ja int32[S], as f64[S]  one slab each, S = sum_b M_b*max_NZ_b,
 *         block b at slot offset off[b]; off int64[nb+1]; pads rewritten
 *         (hip_hll.h).  Row- or col-major inside a block, chosen at upload.
 *
 * Algorithmic bytes per launch (used for roofline.achieved):
 *   CSR   12*NZ + 4*(M+1) + 8*M + 8*N
 *   HLL   12*S + 12*nb + 8*M + 8*N
 */
#ifndef SPMV_ENGINE_H
#define SPMV_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#include "csr.h"
#include "hll.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spmv_launch_opts {
    int waves_per_block; /* 1..16; 0 = process default (set_*_waves_per_block) */
    int group;           /* lanes per row of the sub-wave kernels; 0 = auto */
    int variant;         /* tuning bit-field, 0 = default.  Workgroup -> XCD
                            order of the direct kernels: bit 0 hardware
                            (round-robin) order, bit 1 XCD-contiguous ranges
                            of equal work, grouped runs of 32 workgroups per
                            XCD: bit 2 (HLL kernels 1 / 2) / bit 5 (CSR
                            sub-wave and stream kernels; bit 6: stream kernel
                            in hardware order); none of them: the handle's
                            order (what spmv_*_autotune measured faster;
                            before tuning: grouped from 2M rows up).  CSR
                            stream kernel: bit 4 = 4- / 8-byte loads only.
                            CSR sub-wave kernel: bit 9 = load IRP even when
                            every row has one length (A/B of the constant-
                            row-length path).  Timed loops: bit 29 = the
                            read-modify-write cache flush of rounds 1 / 2.
                            Blocked path: bit 0 launches a chain copy as steps
                            and vice versa, bits 1 / 2 force the tile order
                            (else spmv_panel_opts.tile_order).  Any other bit:
                            -EINVAL, except in an ablations build
                            (spmv_build_flavour) */
    int reserved[5];     /* must be 0 */
} spmv_launch_opts;

/* ---- devices ---- */
int spmv_device_count(void); /* 0 when there is no usable GPU */
int spmv_set_device(int device);
int spmv_get_device(void);   /* current device or negative errno */
/* name (<= len-1 chars), compute units, HBM bytes */
int spmv_device_info(int device, char *name, size_t len, int *compute_units,
                     size_t *hbm_bytes);

/* PCI bus id of a device, "dddd:bb:dd.f" (len >= 16) */
int spmv_device_pci_bus_id(int device, char *buf, size_t len);

/* free / total bytes of HBM on the current device (leak checks) */
int spmv_dev_mem_info(size_t *free_bytes, size_t *total_bytes);

/* ---- raw device memory, for hosts without their own allocator ---- */
int spmv_dev_malloc(void **dptr, size_t bytes);
int spmv_dev_free(void *dptr);
int spmv_dev_memset(void *dptr, int byte, size_t bytes, void *stream);
int spmv_copy_h2d(void *dst_dev, const void *src_host, size_t bytes);
int spmv_copy_d2h(void *dst_host, const void *src_dev, size_t bytes);
int spmv_stream_sync(void *stream);
int spmv_device_sync(void); /* every stream of the current device */
/* GPU timer (reference src/cuda_timer.cu:15-21): events recorded on a stream
 * around whatever the caller enqueues; elapsed_ms waits for `stop` */
int spmv_event_create(void **ev);
int spmv_event_record(void *ev, void *stream);
int spmv_event_elapsed_ms(void *start, void *stop, float *ms);
int spmv_event_destroy(void *ev);
/* Streams, and a hipGraph of whatever is enqueued on one between begin and
 * end: every launch of this API is stream-ordered and capturable (the main
 * kernel, the side launch of long rows / wide hack blocks, the sweep
 * schedule's memset + persistent kernel, the steps schedule's launch
 * sequence), so a solver that iterates y = A x records the launches once and
 * replays them -- x is read where it lives.  Launch each kernel id once
 * eagerly before capturing it (one-off function attributes).  Limits of a
 * replay: INTEGRATION.md "Contracts". */
int spmv_stream_create(void **stream);  /* non-blocking stream */
int spmv_stream_destroy(void *stream);
int spmv_graph_begin_capture(void *stream); /* not the default (NULL) stream */
int spmv_graph_end_capture(void *stream, void **graph_exec);
int spmv_graph_launch(void *graph_exec, void *stream);
int spmv_graph_destroy(void *graph_exec);
/* x[i] = synth_x(seed, first + i) generated on the device */
int spmv_dev_fill_synth(double *d_x, int64_t n, uint64_t seed, int64_t first,
                        void *stream);

/*
 * Extra kernel id of both formats: the 2-D blocked path (no reference
 * counterpart, panels.hip).  The entries are additionally stored bucketed by
 * (row tile, column panel of up to 2^18 columns = 2 MiB of x), 12 B each:
 * a row tile's slice of y accumulates in LDS while the x gathers of a panel
 * stay inside the XCD L2s.  Default schedule "sweep": one persistent launch,
 * every workgroup keeps its tile in LDS across all panels, per-XCD phase
 * counters (bounded waits) keep the workgroups on neighbouring panels.
 * Pays off only when rows reach far beyond an L2 of x (3.6x on config 3 with
 * columns anywhere; slower on matrices with locality -- use
 * spmv_*_autotune); costs 12 B per entry of extra HBM.  Build with
 * spmv_*_build_panels() first (panel_cols = 0: default width), then launch
 * this id (whole matrix only).  opts.variant carries tuning bits
 * (panels.hip), opts.waves_per_block < 8 selects 256-lane workgroups.
 */
#define SPMV_CSR_KERNEL_PANELS 5
#define SPMV_HLL_KERNEL_PANELS 4

/*
 * Explicit build options of the blocked copy (spmv_*_build_panels_opts).
 * Everything a build depends on travels in this struct -- the library reads
 * no environment variable and, apart from the process default schedule of
 * spmv_set_panel_schedule(), keeps no mutable global: two host threads may
 * build and launch different handles concurrently.  Initialise with
 * spmv_panel_opts_default(); 0 means "default" in every field except
 * `struct_size`, `sched` and `sweep_layout` (-1).
 */
typedef struct spmv_panel_opts {
    int struct_size;      /* sizeof(spmv_panel_opts) of the header the caller
                             was compiled with: a build call refuses any other
                             value (-EINVAL), so a caller built against an
                             older, shorter struct fails loudly instead of
                             having the library read past it.  ABI: the struct
                             grew in library versions 0.3 (bucket_order),
                             0.4 (this field, first) and 0.5 (deterministic,
                             last; 0.6: its values 0 / 1 / 2); spmv_version() tells
                             which library is loaded.  Fill the struct with
                             spmv_panel_opts_default() and then set fields */
    int sched;            /* -1 process default, 0 steps, 1 sweep, 2 chain */
    int panel_cols;       /* columns per panel (rounded down to 2^k); 0: 2^18 */
    int tile_rows;        /* rows per tile of steps / chain (32..20448); 0: 4096;
                             the sweep schedule sizes its own tiles */
    int sweep_wgs_per_cu; /* sweep: workgroups sharing a CU's LDS (1..8);
                             0: chosen per matrix */
    int reserve_cus;      /* sweep: compute units left OUT of the persistent
                             grid so that another kernel (RCCL's all-gather of
                             the previous shard) runs beside it; 0: none */
    int lds_min;          /* launch with at least this many bytes of dynamic
                             LDS (caps workgroups per CU); 0: the tile */
    int tile_order;       /* steps / chain, which tile a workgroup runs:
                             0 = grouped (default: groups of 32 consecutive
                             tiles per XCD, groups dealt round-robin --
                             neighbours share an L2 and the chip advances
                             through one region), 1 = hardware order (tile =
                             workgroup index), 2 = XCD-contiguous ranges of
                             equal work.  spmv_*_autotune measures all three */
    int sweep_layout;     /* sweep: 0 = buckets stored tile-major (a workgroup
                             streams one contiguous region), 1 (and -1 =
                             default) = panel-major inside a round (what the
                             chip reads at one time is one compact region).
                             NOTE: the only field whose "default" is -1; a
                             zero-initialised struct asks for layout 0 */
    int bucket_order;     /* steps / chain, the order in which a tile visits
                             its non-empty buckets: 0 = default (ascending
                             panels; when every tile's panels lie within a
                             span K smaller than the number of panels -- a
                             band -- ascending (panel mod K), so that the
                             neighbouring tiles an XCD runs together sit on
                             at most two panels of x at a time), 1 = always
                             ascending panels */
    int deterministic;    /* bitwise-reproducible launches (added in 0.5,
                             last; tri-state since 0.6): 0 = default, 1 = on,
                             2 = off.  Without it the wavefronts of a
                             workgroup add their products into the tile's LDS
                             slice of y with ds_add_f64 as they come: the
                             ORDER of the additions to one row, and so the
                             last bits of y, vary from launch to launch (held
                             to 1e-12 of the row scale).  With it they add
                             chunk by chunk, one wavefront after the other (an
                             LDS turn counter, no barrier): every row's
                             additions happen in one fixed order -- the same
                             bits on every launch of the copy, as the
                             reference's kernels (cuda_hll.cu:49-72, csr.c:
                             201-216) and the nine direct kernels give.
                             DEFAULT: on for sweep layouts, where it is free
                             (+-2 %: the hand-offs hide under the
                             request-bound loop), off for chain / steps
                             (+3..+10 % there, DESIGN.md: opt-in with 1).
                             spmv_*_autotune builds its candidates with the
                             default, so the blocked copy it keeps on a matrix
                             whose columns reach anywhere is reproducible */
} spmv_panel_opts;

/* every field at its default (struct_size set, sched -1, sweep_layout -1,
 * the rest 0) -- the one place the defaults live */
void spmv_panel_opts_default(spmv_panel_opts *opts);

/* ---- CSR handle ---- */
typedef struct spmv_csr_dev spmv_csr_dev;

int spmv_csr_upload(const sparse_csr *A, spmv_csr_dev **out);
/* Build a synthetic matrix (spmv_synth.h) directly in device memory. */
int spmv_csr_generate(int kind, int M, int N, int K, int64_t W, int64_t row0,
                      uint64_t seed, spmv_csr_dev **out);
/* y[0..M) = A * x on `stream`; asynchronous. kernel = 0..4 (hip_csr.h).
 * Launches of one handle must be stream-ordered (the sweep schedule of the
 * blocked path keeps per-handle phase counters); different handles are
 * independent.
 * opts.variant (tuning): see spmv_launch_opts. */
int spmv_csr_launch(const spmv_csr_dev *A, int kernel,
                    const spmv_launch_opts *opts, const double *d_x,
                    double *d_y, void *stream);
/* rows [row_begin, row_end) only; d_y still indexed from row 0.  The stream
 * kernel (4) owns a row-block table of the WHOLE matrix: on a proper sub-range
 * the sub-wave kernel (2) runs instead -- correct, but without the stream
 * kernel's long-row handling; callers that chunk a shard (dist.py, mgpu.hip)
 * therefore launch whole shards when the pick is kernel 4 */
int spmv_csr_launch_rows(const spmv_csr_dev *A, int kernel,
                         const spmv_launch_opts *opts, const double *d_x,
                         double *d_y, int row_begin, int row_end,
                         void *stream);
int spmv_csr_build_panels(spmv_csr_dev *A, int panel_cols);
int spmv_csr_build_panels_opts(spmv_csr_dev *A, const spmv_panel_opts *opts);
/* schedule the NEXT spmv_*_build_panels() calls prepare: 1 = "sweep" (one
 * persistent launch over all panels with phase counters; for rows that reach
 * far beyond an L2 of x; initial value), 2 = "chain" (one launch, a workgroup
 * walks its tile's non-empty buckets; for banded / clustered / skewed
 * matrices), 0 = "steps" (same layout as chain, one launch per non-empty
 * panel step).  -EINVAL otherwise.  spmv_*_autotune tries them all and
 * spmv_*_build_panels_opts / _as name the schedule explicitly. */
int spmv_set_panel_schedule(int sched);
/* geometry of the blocked copy: kernel launches per SpMV (steps), row
 * tiles, column panels, entries kept; -ENOENT when it is not built */
int spmv_csr_panels_info(const spmv_csr_dev *A, int *steps, int *tiles,
                         int *panels, int64_t *entries);
/* blocked copy with the schedule and tile height of `model`'s (shards of one
 * matrix: autotune one, build the others alike) */
int spmv_csr_build_panels_like(spmv_csr_dev *A, const spmv_csr_dev *model);
/* schedule of the blocked copy: 0 steps, 1 sweep, 2 chain; -ENOENT if none */
int spmv_csr_panels_schedule(const spmv_csr_dev *A);
int spmv_csr_panels_tile_rows(const spmv_csr_dev *A); /* -ENOENT if none */
/* one line of text: schedule, tiles x rows, panels x columns, bucket order,
 * launch shape of the blocked copy; -ENOENT if none */
int spmv_csr_panels_describe(const spmv_csr_dev *A, char *buf, size_t len);
/* blocked copy in an explicit schedule (0 steps, 1 sweep, 2 chain) and tile
 * height (0: default; sweep sizes its own tiles): the ranks of a multi-GPU
 * job build what rank 0 tuned */
int spmv_csr_build_panels_as(spmv_csr_dev *A, int panel_cols, int sched,
                             int tile_rows);
/* keep only the blocked copy: frees JA/AS, so the handle costs the HBM of
 * the format it came from (12 B per entry).  Afterwards only the PANELS
 * kernel id runs; the direct kernels, download, conversion and further
 * builds return -ENODATA.  -ENOENT when no blocked copy was built. */
int spmv_csr_release_source(spmv_csr_dev *A);
int spmv_csr_shape(const spmv_csr_dev *A, int *M, int *N, int64_t *NZ);
int64_t spmv_csr_algorithmic_bytes(const spmv_csr_dev *A);
/* download the device arrays into a host CSR (tests; generated matrices) */
int spmv_csr_download(const spmv_csr_dev *A, sparse_csr **out);
void spmv_csr_release(spmv_csr_dev *A);

/* ---- HLL handle ---- */
typedef struct spmv_hll_dev spmv_hll_dev;

int spmv_hll_upload(const sparse_hll *H, int is_col_major, spmv_hll_dev **out);
/* CSR -> HLL conversion on the device (reference hll.c:19-95 semantics,
 * pads already rewritten); the CSR handle stays valid. */
int spmv_hll_from_csr(const spmv_csr_dev *A, int is_col_major,
                      spmv_hll_dev **out);
/* kernel = 0..3 (hip_hll.h); the handle's layout must match the kernel
 * (0,3 row-major; 1,2 col-major) else -EINVAL. */
int spmv_hll_launch(const spmv_hll_dev *H, int kernel,
                    const spmv_launch_opts *opts, const double *d_x,
                    double *d_y, void *stream);
/* hack blocks [blk_begin, blk_end) only */
int spmv_hll_launch_blocks(const spmv_hll_dev *H, int kernel,
                           const spmv_launch_opts *opts, const double *d_x,
                           double *d_y, int blk_begin, int blk_end,
                           void *stream);
/* HLL source: pad slots (JA == -1 on the host; remembered in a bitmap when
 * the pads are rewritten at upload) are dropped; explicit zeros are kept,
 * as from a CSR source */
int spmv_hll_build_panels(spmv_hll_dev *H, int panel_cols);
int spmv_hll_build_panels_opts(spmv_hll_dev *H, const spmv_panel_opts *opts);
int spmv_hll_panels_info(const spmv_hll_dev *H, int *steps, int *tiles,
                         int *panels, int64_t *entries);
int spmv_hll_build_panels_like(spmv_hll_dev *H, const spmv_hll_dev *model);
int spmv_hll_panels_schedule(const spmv_hll_dev *H);
/* the blocked copy's layout as build options + the launch's waves hint (set
 * o->struct_size first; -ENOENT: no copy): spmv_*_build_panels_opts(o) and
 * spmv_*_panels_set_waves(waves) on a handle of the same matrix rebuild
 * exactly what spmv_*_autotune settled on */
int spmv_csr_panels_layout(const spmv_csr_dev *A, spmv_panel_opts *o, int *waves);
int spmv_hll_panels_layout(const spmv_hll_dev *H, spmv_panel_opts *o, int *waves);
int spmv_csr_panels_set_waves(spmv_csr_dev *A, int waves); /* 0..16 */
int spmv_hll_panels_set_waves(spmv_hll_dev *H, int waves);
int spmv_hll_panels_tile_rows(const spmv_hll_dev *H);
int spmv_hll_panels_describe(const spmv_hll_dev *H, char *buf, size_t len);
int spmv_hll_build_panels_as(spmv_hll_dev *H, int panel_cols, int sched,
                             int tile_rows);
int spmv_hll_release_source(spmv_hll_dev *H);
int spmv_hll_shape(const spmv_hll_dev *H, int *M, int *N, int64_t *NZ,
                   int *num_blocks, int64_t *slots, int *is_col_major);
int64_t spmv_hll_algorithmic_bytes(const spmv_hll_dev *H);
/* bytes one launch of `kernel` must move: spmv_hll_algorithmic_bytes (12 per
 * STORED slot) for the direct kernels; for SPMV_HLL_KERNEL_PANELS, whose
 * copy stores no padding, 12 per true entry (+ 12 nb + 8 M + 8 N as before).
 * Equal when the format pads nothing (the headline matrix). */
int64_t spmv_hll_kernel_bytes(const spmv_hll_dev *H, int kernel);
void spmv_hll_release(spmv_hll_dev *H);

/*
 * Timed loops: `warmup` untimed launches, then `iters` launches each
 * bracketed by its own hipEvent pair on `stream`; ms_each[iters] receives
 * the per-launch kernel times.  flush_bytes > 0 streams a scratch buffer of
 * that size between iterations (outside the timed region) so that matrices
 * smaller than the 256 MiB Infinity Cache are read from HBM.  The sweep only
 * READS the scratch buffer (no dirty lines whose write-back would overlap the
 * timed launch); opts.variant bit 29 selects the read-modify-write sweep of
 * rounds 1 / 2 for comparison.
 */
int spmv_csr_time(const spmv_csr_dev *A, int kernel,
                  const spmv_launch_opts *opts, const double *d_x, double *d_y,
                  int warmup, int iters, size_t flush_bytes, double *ms_each,
                  void *stream);
int spmv_hll_time(const spmv_hll_dev *H, int kernel,
                  const spmv_launch_opts *opts, const double *d_x, double *d_y,
                  int warmup, int iters, size_t flush_bytes, double *ms_each,
                  void *stream);

/*
 * Pick the fastest kernel for this matrix by measurement (5 launches each):
 * the coalesced kernels of the handle's layout and, when allow_panels != 0
 * and those run well below the stream rate, the 2-D blocked path (built on
 * demand, released again if it loses).  d_y is scratch.
 */
int spmv_csr_autotune(spmv_csr_dev *A, const double *d_x, double *d_y,
                      int allow_panels, int *best_kernel, double *best_ms);
int spmv_hll_autotune(spmv_hll_dev *H, const double *d_x, double *d_y,
                      int allow_panels, int *best_kernel, double *best_ms);

/* per-kernel medians (ms) of the last spmv_*_autotune on this handle, kernel
 * ids 0 .. n-1 (0.0: not a candidate for this matrix), and what the selector
 * did: one text line per phase with host-clock seconds -- direct kernels,
 * every blocked candidate (build s, configurations timed, s, best ms), total
 * (-ENOENT before any autotune) */
int spmv_csr_tune_times(const spmv_csr_dev *A, double *ms, int n);
int spmv_hll_tune_times(const spmv_hll_dev *H, double *ms, int n);
int spmv_csr_tune_log(const spmv_csr_dev *A, char *buf, size_t len);
int spmv_hll_tune_log(const spmv_hll_dev *H, char *buf, size_t len);

/*
 * Dead-handle contract.  Every entry point that takes a handle checks it
 * first: NULL -> -EINVAL, a pointer that is not (or no longer) a live handle
 * -> -EBADF, before anything is allocated or dereferenced
 * (spmv_*_algorithmic_bytes return the code as a negative byte count).
 * spmv_*_release() of such a pointer is ignored AND counted:
 * spmv_ignored_releases() returns the count, spmv_set_debug(1) adds one line
 * on stderr per ignored release (the library reads no environment variable;
 * the Python binding passes SPMV_DEBUG on).  spmv_live_handles(): handles
 * created and not yet released (CSR + HLL, one-shot calls included while they
 * run).
 *
 * Address reuse: a released handle's address may be handed out again by the
 * allocator.  Every handle therefore carries a process-wide generation
 * (1, 2, 3, ... never reused; spmv_handle_generation, 0 for a pointer that is
 * not live), and spmv_*_release_checked(h, generation) releases only the
 * handle that was created with that generation -- a stale wrapper object
 * (finaliser, atexit sweep) cannot release a newer handle that happens to live
 * at the old address.  Bindings with finalisers should use the checked form.
 */
int spmv_live_handles(void);
long spmv_ignored_releases(void);
void spmv_set_debug(int on);
uint64_t spmv_handle_generation(const void *handle);
void spmv_csr_release_checked(spmv_csr_dev *A, uint64_t generation);
void spmv_hll_release_checked(spmv_hll_dev *H, uint64_t generation);

/* Library self-description: "spmv_scpa_amd <version> gfx950". */
const char *spmv_version(void);
/* TEST HOOKS: leave in every last-arriver counter of the handle (long rows,
 * wide hack blocks, the long rows beside a blocked copy) what a launch that
 * never completed would have left.  Later launches must give the right y all
 * the same: the counters carry the launch's number and an arrival that finds
 * another number starts from zero.  Returns the counters touched. */
int spmv_csr_debug_stale_arrivals(spmv_csr_dev *A);
int spmv_hll_debug_stale_arrivals(spmv_hll_dev *H);

/* HIP_VERSION (major * 10^7 + minor * 10^5 + patch) of the headers the library
 * was built with / of the runtime the process has bound it to (-EIO when the
 * runtime does not answer).  Majors must agree. */
int spmv_hip_build_version(void);
int spmv_hip_runtime_version(void);
/* "product", or "ablations": built with -DSPMV_ABLATIONS (`make abl` ->
 * lib/libspmv_scpa_amd_abl.so), the only flavour that accepts the experiment
 * bits of spmv_launch_opts.variant (timing ablations, some of which compute a
 * WRONG y by design; pipeline depths, phase lags, group sizes).  The product
 * library answers -EINVAL to any bit that is not documented above. */
const char *spmv_build_flavour(void);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_ENGINE_H */

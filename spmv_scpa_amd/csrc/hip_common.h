/*
 * hip_common.h -- internals shared by the HIP translation units
 * (engine.hip, csr_kernels.hip, hll_kernels.hip).  Not installed.
 */
#ifndef SPMV_HIP_COMMON_H
#define SPMV_HIP_COMMON_H

#include <errno.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spmv_engine.h"

#define WAVE 64 /* gfx950 wavefront */
/* MI355X: 8 accelerator dies, workgroups are dealt to them round-robin.  The
 * library is built for gfx950 only (Makefile) and refuses other devices
 * (spmv_set_device / panels_build check the arch name), so this is a
 * build-time constant rather than a queried property -- HIP exposes no XCD
 * count. */
#define NUM_XCD 8

/* HIP status -> negative errno (the host API's error convention) */
static inline int hip_errno(hipError_t e) {
    switch (e) {
    case hipSuccess:
        return 0;
    case hipErrorNoDevice:
    case hipErrorInvalidDevice:
    case hipErrorInsufficientDriver:
    case hipErrorNotInitialized:
        return -ENODEV;
    case hipErrorOutOfMemory:
        return -ENOMEM;
    case hipErrorInvalidValue:
    case hipErrorInvalidConfiguration:
        return -EINVAL;
    default:
        return -EIO;
    }
}

#define HIP_TRY(call)                                                         \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            rc = hip_errno(e_);                                               \
            goto fail;                                                        \
        }                                                                     \
    } while (0)

#define HIP_RET(call)                                                         \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess)                                                 \
            return hip_errno(e_);                                             \
    } while (0)

/* variant bits the launchers ignore (they belong to the timed loops) */
#define SPMV_VARIANT_TIMING_BITS (1 << 29)

/* nnz budget of one workgroup of the CSR stream kernel */
#define STREAM_NNZ 2048
#define STREAM_THREADS 256
/* rows a range may hold: matrices with 2-5 entries per row (web / road /
 * co-purchase graphs) fill the entry budget only with several rows per lane */
#define STREAM_ROWS 1024
/* ranges whose longest row is at most this use the transposed form */
#define STREAM_ROW_T 48
/* hack blocks wider than HLL_WIDE columns go to k_hll_wide in segments of
 * HLL_WSEG columns (hll_kernels.hip) */
#define HLL_WIDE 512
#ifndef HLL_WSEG
#define HLL_WSEG 256      /* narrowest segment */
#endif
#ifndef HLL_WSEG_MAX
#define HLL_WSEG_MAX 512  /* most segments per block: a hub block of 5 x 10^5
                             columns is then cut into ~1000-column segments
                             (the block's last segment adds up one partial
                             sum per segment and row) */
#endif
/* a row beyond STREAM_LONG_ROW entries is cut into segments of
 * STREAM_SEG entries, one workgroup each (one lane team walking a row of
 * 10^5 entries is the whole launch otherwise: dc1-class matrices) */
#define STREAM_LONG_ROW 8192
#ifndef STREAM_SEG
#define STREAM_SEG 2048
#endif

struct spmv_panels; /* panels.hip */

/* contiguous ranges of a launch's work items, one per XCD, holding about
 * equal numbers of ENTRIES: workgroup b (dealt round-robin to the XCDs) runs
 * item first[b % NUM_XCD] + b / NUM_XCD, or nothing when that lies beyond
 * its XCD's range; the launch has NUM_XCD * (longest range) workgroups.
 * Neighbouring items share their window of x, so they meet in one L2 --
 * and no XCD idles when the rows are denser in one part of the matrix. */
struct xcd_ranges {
    int first[NUM_XCD + 1];
};

struct spmv_csr_dev {
    int M, N;
    int64_t NZ;
    int device;
    int *irp;   /* [M+1] */
    int *ja;    /* [NZ]  */
    double *as; /* [NZ]  */
    /* stream kernel: workgroup k owns rows [rowblk[2k], rowblk[2k+2]),
     * entries [rowblk[2k+1], rowblk[2k+3]) */
    int *rowblk;
    int n_rowblk;
    unsigned char *rowblk_mode; /* per range: 0 transposed, 1 cooperative,
                                   2 segment of a long row */
    double *seg_partial; /* [n_rowblk] partial sum of a segment's range */
    unsigned long long *seg_count; /* [n_rowblk] (launch epoch << 32 |
                            arrivals), at a long row's first range
                            (epoch_arrive below); both NULL when no row is
                            that long */
    unsigned launch_epoch; /* number of the last launch that counted arrivals */
    int *long_rb;        /* [n_long_rb] indices of the ranges that hold (a
                            segment of) ONE row of more than STREAM_NNZ
                            entries: kernels 0-3 leave such rows to a second
                            launch over exactly these ranges */
    int n_long_rb;
    int max_row_len;
    int uniform_len; /* > 0: EVERY row holds exactly this many entries (banded
                        and fixed-degree matrices: IRP[r] = r * uniform_len),
                        so the sub-wave kernel need not wait for IRP before it
                        can fetch JA / AS; 0: row lengths vary */
    int order; /* sub-wave kernel, workgroup order: 0 hardware, 1 XCD-contiguous
                  equal ranges, 2 grouped runs of 32 workgroups per XCD
                  (spmv_csr_autotune measures all three) */
    int stream_grouped; /* stream kernel: ranges in grouped runs instead of
                           hardware order (spmv_csr_autotune measures both) */
    spmv_panels *panels; /* optional column-panel copy (kernel 5) */
    double tune_ms[8];   /* last spmv_csr_autotune: best median per kernel id
                            (0: not timed) */
    char *tune_log;      /* ... and what it did, one line per phase (malloc) */
};

struct spmv_hll_dev {
    int M, N;
    int64_t NZ;
    int device;
    int nb;        /* hack blocks */
    int col_major; /* layout inside a block */
    int64_t slots; /* S */
    int max_width; /* largest max_NZ */
    int *ja;       /* [S] pads already rewritten */
    double *as;    /* [S] */
    int64_t *off;  /* [nb+1] slot offset of each block */
    int order; /* kernels 1 / 2, which blocks a workgroup runs: 0 = hardware
                  order (eight XCDs advancing through ONE region of the
                  slabs: 3-8 % faster on banded matrices than eight contiguous
                  regions), 1 = the XCD ranges below, 2 = grouped (runs of 32
                  workgroups per XCD, runs round-robin: region locality AND
                  neighbours in one L2 -- banded 10M x 32: 85 % of 8 TB/s);
                  spmv_hll_autotune measures all three */
    xcd_ranges xcd_blk; /* hack-block ranges per XCD holding ~1/8 of the SLOTS
                           each (even boundaries: a wavefront owns a pair) */
    unsigned *padmask; /* [(S+31)/32] bit t set: slot t was a pad (JA == -1)
                          before the rewrite; read only when the blocked copy
                          is built */
    spmv_panels *panels; /* optional column-panel copy (kernel 4) */
    /* WIDE hack blocks (a block is as wide as its longest row: one hub row
     * of 10^5 entries makes 32 lanes walk 10^5 columns, 10-60 ms; even 4096
     * columns are a 0.2-0.5 ms walk of dependent steps).  Blocks
     * wider than HLL_WIDE columns are skipped by kernels 0-3 and summed by a
     * second launch, one workgroup per segment of HLL_WSEG columns
     * (k_hll_wide; deterministic last-arriver reduction per block) */
    int4 *wide_seg;    /* [n_wide_seg] (block, segment width, segment, segments) */
    int n_wide_seg;
    double *wide_part; /* [n_wide_seg * 32] partial row sums */
    unsigned long long *wide_cnt; /* [n_wide_seg] (launch epoch << 32 |
                          arrivals), at a block's first segment */
    unsigned launch_epoch;
    double tune_ms[8];   /* last spmv_hll_autotune: best median per kernel id */
    char *tune_log;
};

/* The next launch number of a handle, never 0 (a zeroed counter belongs to
 * no launch).  The handle is `const` to the launch path but this one field
 * moves; launches of ONE handle are stream-ordered by contract, the atomic
 * only keeps two host threads that break it from tearing the number. */
static inline unsigned next_launch_epoch(const unsigned *field) {
    unsigned *f = const_cast<unsigned *>(field);
    unsigned e = __atomic_add_fetch(f, 1u, __ATOMIC_RELAXED);
    if (e == 0)
        e = __atomic_add_fetch(f, 1u, __ATOMIC_RELAXED);
    return e;
}

#if defined(__HIPCC__)
/*
 * Arrival at a last-arriver counter.  The counter holds (launch epoch << 32 |
 * arrivals of that launch).  Fast path: ONE fetch-and-add that finds the
 * launch's own epoch -- what every arrival of an ordinary launch takes (the
 * last arriver of the launch before left (epoch + 1, 0), and a handle's
 * launches are numbered consecutively).  An arrival that finds ANOTHER epoch
 * -- a launch that never completed left its count behind (a fault, a process
 * killed between the main kernel and its side launch: that used to make every
 * later launch of the handle reduce too early, a wrong y[row] with no error,
 * ADVICE r04), or the launch is a REPLAY of a captured hipGraph, whose kernel
 * arguments are frozen -- resets the word to (epoch, 1) with a compare-and-
 * swap, or, when another arrival has done so meanwhile, adds itself to it:
 * every arrival is counted exactly once on the launch's own word (the add it
 * spent on the foreign word counted nothing).  A compare-and-swap loop for
 * EVERY arrival was measured first: 128 segments of a hub row meeting on one
 * word went from 7 to 90 us (hub 1M: 0.030 -> 0.116 ms).  Returns this
 * arrival's number, 1 .. n; the caller that gets n is the last and calls
 * epoch_rearm().  Agent scope: the arrivals come from different XCDs.
 */
__device__ __forceinline__ unsigned epoch_arrive(unsigned long long *cnt,
                                                 unsigned epoch) {
    unsigned long long old = __hip_atomic_fetch_add(
        cnt, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(old >> 32) == epoch)
        return (unsigned)old + 1u;
    old = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        if ((unsigned)(old >> 32) == epoch) /* reset by another arrival */
            return (unsigned)__hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_ACQ_REL,
                                                    __HIP_MEMORY_SCOPE_AGENT) +
                   1u;
        const unsigned long long first = ((unsigned long long)epoch << 32) | 1ull;
        if (__hip_atomic_compare_exchange_strong(cnt, &old, first,
                                                 __ATOMIC_ACQ_REL,
                                                 __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT))
            return 1u;
    }
}

/* the last arriver leaves (epoch + 1, 0 arrivals): the word the handle's next
 * launch expects (see epoch_arrive) */
__device__ __forceinline__ void epoch_rearm(unsigned long long *cnt,
                                            unsigned epoch) {
    __hip_atomic_store(cnt, (unsigned long long)(epoch + 1u) << 32,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/* part[first * stride], part[(first + step) * stride], ... (indices below n)
 * added up in THAT order, U loads in flight at a time.  The partial sums were
 * written by other workgroups, on other XCDs: agent-scope atomic loads, each a
 * round trip to the L2 / fabric -- one at a time, the 64 loads per lane of a
 * 512-segment hack block were most of k_hll_wide's launch. */
template <int U>
__device__ __forceinline__ double ordered_partial_sum(const double *part, int n,
                                                      int first, int step,
                                                      size_t stride) {
    double s = 0.0;
    for (int j = first; j < n; j += step * U) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            v[u] = j + u * step < n
                       ? __hip_atomic_load(part + (size_t)(j + u * step) * stride,
                                           __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT)
                       : 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (j + u * step < n)
                s += v[u];
    }
    return s;
}

/* Sum part[0 .. n) in a FIXED order with one wavefront (all 64 lanes call):
 * lane l adds part[l], part[l + 64], ... in index order, then a shuffle tree.
 * Total in lane 0.  (One lane adding 10^2..10^3 partials one dependent load
 * at a time was a third of the hub matrices' launch.) */
__device__ __forceinline__ double wave_ordered_sum(const double *part, int n,
                                                   int lane) {
    double s = ordered_partial_sum<4>(part, n, lane, WAVE, 1);
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1)
        s += __shfl_down(s, d, WAVE);
    return s;
}

/* Columns of hack block b (`rows` = its row count).  The slot count is 64-bit:
 * a hub block of 2^27 columns holds 2^32 slots, and `(unsigned)(off[b + 1] -
 * off[b])` -- a 32-bit division is cheaper -- silently halved such a block */
__device__ __forceinline__ int hack_block_width(const int64_t *__restrict__ off,
                                                int b, int rows) {
    const uint64_t n = (uint64_t)(off[b + 1] - off[b]);
    return rows == HACK_SIZE ? (int)(n >> 5) : (int)(n / (unsigned)rows);
}

/* Thread `tid` of NT sums as[k] * x[ja[k]] over k = beg + tid, + NT, ... < end
 * in THAT order (one accumulator: the bits do not depend on U), with the loads
 * of U entries in flight at a time.  The plain loop is two dependent memory
 * round trips per entry -- 4 us per 8 entries on a busy chip -- and a long
 * row's segment is 4-32 entries per thread: the side launches of the long
 * rows were latency, not bandwidth.  Entries stream past (non-temporal). */
template <int NT, int U>
__device__ __forceinline__ double strided_dot(const int *__restrict__ ja,
                                              const double *__restrict__ as,
                                              const double *__restrict__ x,
                                              int beg, int end, int tid) {
    double acc = 0.0;
    /* indices relative to `beg`: `end` may sit next to INT32_MAX (the entry
     * count's limit), beg + k + U * NT must not be formed in 32 bits */
    ja += beg;
    as += beg;
    end -= beg;
    int k = tid;
    for (; k + (U - 1) * NT < end; k += U * NT) {
        int c[U];
        double v[U], xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            c[u] = __builtin_nontemporal_load(ja + k + u * NT);
            v[u] = __builtin_nontemporal_load(as + k + u * NT);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            xv[u] = x[c[u]];
#pragma unroll
        for (int u = 0; u < U; ++u)
            acc += v[u] * xv[u];
    }
    if (k < end) { /* fewer than U left: one predicated batch */
        int c[U];
        double v[U], xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool on = k + u * NT < end;
            c[u] = on ? __builtin_nontemporal_load(ja + k + u * NT) : -1;
            v[u] = on ? __builtin_nontemporal_load(as + k + u * NT) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            xv[u] = c[u] >= 0 ? x[c[u]] : 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (c[u] >= 0)
                acc += v[u] * xv[u];
    }
    return acc;
}
#endif

/* clamp the launch knob: waves per workgroup */
static inline int pick_waves(const spmv_launch_opts *o, int dflt) {
    int w = (o && o->waves_per_block > 0) ? o->waves_per_block : dflt;
    if (w < 1)
        w = 1;
    if (w > 16)
        w = 16;
    return w;
}

/* kernel launchers (csr_kernels.hip / hll_kernels.hip) */
int csr_launch_kernel(const spmv_csr_dev *A, int kernel, int waves, int group,
                      int variant, const double *x, double *y, int r0, int r1,
                      hipStream_t s);
int hll_launch_kernel(const spmv_hll_dev *H, int kernel, int waves,
                      int variant, const double *x, double *y, int b0, int b1,
                      hipStream_t s);
int hll_fix_pads_dev(spmv_hll_dev *H, hipStream_t s);

/* column-panel path (panels.hip) */
int panels_from_csr(const spmv_csr_dev *A, int panel_cols, int sched,
                    int tile_rows, spmv_panels **out);
int panels_from_hll(const spmv_hll_dev *H, int panel_cols, int sched,
                    int tile_rows, spmv_panels **out);
int panels_from_csr_opts(const spmv_csr_dev *A, const spmv_panel_opts *o,
                         spmv_panels **out);
int panels_from_hll_opts(const spmv_hll_dev *H, const spmv_panel_opts *o,
                         spmv_panels **out);
void panels_get_opts(const spmv_panels *P, spmv_panel_opts *o);
int panels_is_sweep(const spmv_panels *P);
int panels_tile_rows(const spmv_panels *P);
int panels_is_chain(const spmv_panels *P);
void panels_set_chain(spmv_panels *P, int chain);
void panels_set_waves(spmv_panels *P, int waves);
int panels_debug_stale_arrivals(spmv_panels *P);
void panels_set_order(spmv_panels *P, int order);
int panels_waves(const spmv_panels *P);
int panels_launch(const spmv_panels *P, int M, int waves, int variant,
                  const double *x, double *y, hipStream_t s);
void panels_free(spmv_panels *p);
int64_t panels_nnz(const spmv_panels *P);
int panels_count(const spmv_panels *P);
int panels_steps(const spmv_panels *P);
int panels_tiles(const spmv_panels *P);
int panels_balanced_tile_rows(int M, int max_rows);
int panels_describe(const spmv_panels *P, char *buf, size_t len);
const char *panels_last_build_phases(void);
void panels_pool_begin(void); /* block reuse across the builds of one */
void panels_pool_end(void);   /* selector run (panels.hip, build_pool)  */

extern int g_csr_waves; /* process defaults behind set_*_waves_per_block */
extern int g_hll_waves;

#endif /* SPMV_HIP_COMMON_H */

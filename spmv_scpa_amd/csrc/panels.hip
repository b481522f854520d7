/*
 * panels.hip -- 2-D blocked path: row tiles accumulate in LDS, column panels
 * of x stay in the XCD L2s (no reference counterpart; extra kernel id of
 * both formats, spmv_engine.h).
 *
 * Why: on MI355X an 8-byte gather of x that misses L1 moves a whole 128-byte
 * line.  From the XCD's 4 MiB L2 that runs at ~265 G gathers/s; from beyond
 * L2 (Infinity Cache and HBM alike) the fabric caps it at ~55 G gathers/s
 * (profiles/r01_microbench_mi355x.txt).  A 10M x 10M matrix with columns
 * anywhere spends 5.8 ms on its 320 M gathers whatever the kernel does.
 *
 * What: at upload the entries are bucketed by (row tile, column panel): a
 * tile is a range of consecutive rows whose slice of y fits LDS, a panel is
 * up to 2^18 columns (2 MiB of x).  Entries are stored tile-major and,
 * inside a tile, in panel order, 12 B each: one 32-bit word (row-in-tile <<
 * shift | column-in-panel) and the fp64 value; every bucket is column-sorted
 * and laid out in blocks of 256 entries (block_ent_slot).  Products are
 * added into the LDS tile with the hardware LDS fp64 atomic (ds_add_f64): no
 * segmented reduction, no row-length sensitivity.
 *
 * Schedule "steps" (tiles of 32 .. 16384 rows): ONE LAUNCH PER STEP; in step s
 * workgroup t adds the s-th non-empty bucket of tile t into its y slice
 * (coalesced read-modify-write through LDS; a tile has one owner per launch,
 * launches are stream-ordered, step 0 starts from zero).  Every CU gathers
 * from the same or a neighbouring panel by construction.  Cost: y is re-read
 * and re-written once per further panel that touches the tile (6.2 of the
 * 11.6 GB per SpMV on config 3) -- nothing for a matrix whose tiles stay
 * inside one panel, which is where this schedule wins: with the buckets
 * column-sorted a banded or skewed matrix turns into a pure stream.
 *
 * Schedule "sweep" (one workgroup per CU with a tile of up to 20448 rows,
 * or two with up to 10208 rows each, all resident): ONE persistent launch; a workgroup keeps its tile of y in LDS while it
 * walks the panels in order and writes y once.  The workgroups of an XCD are
 * kept within `lag` panels of each other by per-XCD phase counters (one
 * agent-scope add per workgroup and phase, relaxed polls): the wait is only
 * there for L2 locality -- it is BOUNDED, and falling through it costs
 * speed, never correctness, so the kernel cannot hang on a chip that does
 * not hold the whole grid.
 *
 * Schedule "chain" (same layout as steps): ONE launch, workgroup t walks
 * all non-empty buckets of tile t with its slice of y in LDS and writes it
 * once; no phase control -- neighbouring tiles stay on neighbouring panels
 * by themselves when a tile has only a handful of buckets.  The fastest form
 * for banded / clustered / skewed matrices (k_tiles_chain).
 *
 * The path is opt-in (12 B per entry of extra HBM) and the autotuner keeps
 * it, in the schedule that measures faster, only when it beats the direct
 * kernels.
 *
 * Summation order inside a row depends on LDS atomic arrival order: results
 * are reproducible to rounding (tests hold them to 1e-12 of the row scale),
 * not bitwise.
 */
#include <algorithm>
#include <atomic>
#include <utility>
#include <hipcub/hipcub.hpp>
#include <stdio.h>
#include <string.h>
#include <vector>

#include <time.h>

#include "hip_common.h"

#define TILE_ROWS_STEPS 4096 /* "steps" schedule default: 32 KiB of LDS */
#define SWEEP_WG_PER_CU 2
#define SWEEP_SPIN_MAX 512   /* polls (~1 us each) before a wait gives up:
                                perf only, see phase_wait */
#define CNT_STRIDE 32        /* one phase counter per 128-B line */
#define SWEEP_TAIL 16384     /* zero slots behind the entries: >= 3 chunks */
#define BIG_LDS_BYTES (160 * 1024 - 256) /* most dynamic LDS a launch asks for */

struct spmv_panels {
    int N;           /* columns */
    int shift;       /* log2(columns per panel) */
    int panels;      /* column panels */
    int tile_rows;   /* rows per tile (multiple of 32) */
    int tiles;       /* row tiles */
    int sweep;       /* built for the persistent schedule */
    int chain;       /* steps layout, launched as one chain launch */
    int waves_hint;  /* wavefronts per workgroup when the caller passes 0
                        (set by the autotuner; 0 = the built-in heuristic) */
    int grid;        /* sweep: workgroups of the launch */
    int wgs_per_cu;  /* sweep: workgroups sharing a CU's LDS */
    int reserve_cus; /* sweep: CUs left out of the grid (spmv_panel_opts) */
    int pmajor;      /* sweep: buckets stored panel-major inside a round */
    int lds_min;     /* launch with at least this much dynamic LDS; 0: tile */
    int64_t nnz;     /* entries kept */
    int64_t total;   /* slots of ENT/VAL in use (bucket padding included) */
    unsigned *ent;   /* [nnz] row-in-tile << shift | column-in-panel */
    double *val;     /* [nnz] */
    int64_t *bptr;   /* DEVICE [tiles*panels+1] start of bucket (tile, panel);
                        multiples of 256 slots (one wavefront's block) */
    int *blen;       /* DEVICE [tiles*panels] entries of the bucket (the slots
                        up to the next start are padding, never read as data) */
    int64_t *cb;     /* DEVICE [tiles*panels*2] (begin,end) of the s-th NON-EMPTY
                        bucket of each tile, tile-major */
    int *cpanel;     /* DEVICE [tiles*panels] panel of that bucket */
    int *nbk;        /* DEVICE [tiles] non-empty buckets per tile */
    int max_nbk;     /* launches needed = max over tiles */
    int span;        /* steps / chain: widest run of panels a tile touches */
    int residue;     /* steps / chain: buckets listed in residue order */
    int bucket_order; /* spmv_panel_opts.bucket_order the copy was built with */
    int det;          /* spmv_panel_opts.deterministic: ordered LDS additions */
    /* steps / chain: XCD k runs the CONTIGUOUS tile range
     * [xcd_first[k], xcd_first[k+1]) -- neighbouring tiles share their x
     * window, so they should meet in one L2 -- and the ranges hold about
     * equal numbers of ENTRIES, not of tiles: with equal tile counts a
     * matrix whose rows are dense in one half (the nlpkkt160-shaped KKT
     * matrix: 42 entries per state row, 15 per constraint row) leaves half
     * of the XCDs idle while the others finish (0.62 ms; balanced: see
     * DESIGN.md) */
    int xcd_first[NUM_XCD + 1];
    int xcd_max;     /* longest range: the launch has NUM_XCD * xcd_max groups */
    int order;       /* steps / chain, which tile a workgroup runs
                        (spmv_panel_opts.tile_order; the selector measures):
                        0 grouped, 1 hardware order, 2 XCD-contiguous ranges */
    int *phase_cnt;  /* DEVICE sweep: [NUM_XCD][rounds*panels] arrival counters */
    size_t phase_cnt_bytes;
    /* LONG ROWS BESIDE THE COPY (round 4).  A row of more than PANELS_LONG_ROW
     * (16384) entries is left out of the buckets: its products would all land on ONE
     * accumulator of its tile, and ds_add_f64 from 64 lanes to one address
     * runs one lane at a time (a hub row of 131072 entries made its tile the
     * whole launch: 0.18 ms for a matrix that streams in 0.02).  Such rows
     * are kept as a small CSR of their own, cut into segments of
     * PANELS_LONG_SEG entries, and a second launch on the same stream sums
     * each segment with a workgroup (k_long_rows; deterministic last-arriver
     * reduction, as the CSR stream kernel's mode 2) and overwrites y[row] --
     * the tile kernels wrote 0 there. */
    int nlong;        /* rows kept beside the copy */
    int nlong_seg;    /* their segments = workgroups of the second launch */
    int64_t long_nnz; /* their entries */
    int *long_row;    /* DEVICE [nlong] row index, ascending */
    int *long_ptr;    /* DEVICE [nlong+1] first entry of each in long_ja/as */
    int *long_ja;     /* DEVICE [long_nnz] */
    double *long_as;  /* DEVICE [long_nnz] */
    int2 *long_seg;   /* DEVICE [nlong_seg] (index into long_row, first entry) */
    int *long_seg0;   /* DEVICE [nlong] first segment of each row */
    double *long_part; /* DEVICE [nlong_seg] partial sums */
    unsigned long long *long_cnt; /* DEVICE [nlong] (launch epoch << 32 |
                         arrivals): epoch_arrive, hip_common.h */
    unsigned launch_epoch;
};
/* threshold: the second launch costs ~8 us; a row inside the copy costs its
 * tile ~1.3 ns per entry of serialised accumulation -- power-law 1M x 3 with
 * one row of 8291 entries: 0.043 ms inside, 0.052 beside; a row of 30 000+
 * entries is worth the launch (4M rows: 0.155 -> 0.127 ms) */
#define PANELS_LONG_ROW 16384
#ifndef PANELS_LONG_SEG
#define PANELS_LONG_SEG 1024
#endif
#define PANELS_LONG_MAX 4096 /* more such rows than this: keep them inside */

static void big_free(void *p); /* block pool of a selector run, below */

void panels_free(spmv_panels *p) {
    if (!p)
        return;
    big_free(p->ent);
    big_free(p->val);
    (void)hipFree(p->cpanel);
    (void)hipFree(p->phase_cnt);
    (void)hipFree(p->bptr);
    (void)hipFree(p->blen);
    (void)hipFree(p->cb);
    (void)hipFree(p->nbk);
    (void)hipFree(p->long_row);
    (void)hipFree(p->long_ptr);
    (void)hipFree(p->long_ja);
    (void)hipFree(p->long_as);
    (void)hipFree(p->long_seg);
    (void)hipFree(p->long_seg0);
    (void)hipFree(p->long_part);
    (void)hipFree(p->long_cnt);
    free(p);
}

/* ---- keys: (tile * panels + panel) << shift | column-in-panel: buckets in
 * order and, inside a bucket, ascending columns, so the lanes of a gather
 * instruction ask for neighbouring lines of x and entries that share a line
 * share the request.  dropped slots: bucket id nbuckets (sorts last) ---- */
/* bucket id of (tile, panel): tile-major, or -- sweep layout 1, pm_grid > 0 --
 * panel-major inside a round of pm_grid tiles: ((round * panels + panel) *
 * pm_grid + tile % pm_grid), so that what the chip reads at one time (every
 * workgroup's bucket of the same panel) is one compact region of ENT / VAL */
__host__ __device__ __forceinline__ uint64_t bucket_id(uint64_t tile,
                                                       uint64_t panel,
                                                       uint64_t panels,
                                                       uint64_t pm_grid) {
    return pm_grid ? ((tile / pm_grid) * panels + panel) * pm_grid + tile % pm_grid
                   : tile * panels + panel;
}

/* is `row` one of the ascending `rows[0..n)`?  (the long rows: a handful) */
__device__ __forceinline__ bool in_sorted(const int *__restrict__ rows, int n,
                                          int row) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (rows[mid] < row)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo < n && rows[lo] == row;
}

/*
 * Sort key of an entry: [bucket id | column in panel (shift bits) | row in
 * tile (rbits bits)], value = the entry's fp64 value.  The radix sort runs
 * over the bits ABOVE the row field only (begin_bit = rbits): buckets in id
 * order, column-sorted inside, ties in source order (the sort is stable, the
 * source row-major: a deterministic layout) -- and the row rides along, so
 * that the sorted (key, value) pairs are everything the copy holds: placing
 * them (k_place) is a pure stream.  Until round 5 the sort carried the
 * source INDEX and a gather kernel fetched ja / as at sorted, i.e. random,
 * positions and searched the row (a bisection per entry): 80.6 GB moved to
 * write a 3.84 GB copy (profiles/r04_wn_hll_tile_panels.md).
 */
__device__ __forceinline__ uint64_t entry_key(uint64_t bucket, unsigned col_low,
                                              unsigned row_in_tile, int shift,
                                              int rbits) {
    return (((bucket << shift) | col_low) << rbits) | row_in_tile;
}

__global__ void k_keys_from_csr(int M, int tile_rows, int panels, int shift,
                                int rbits, int pm_grid, uint64_t nbuckets,
                                const int *__restrict__ long_row, int nlong,
                                const int *__restrict__ irp,
                                const int *__restrict__ ja,
                                const double *__restrict__ as, uint64_t *key,
                                double *val) {
    /* 8 lanes per row keep the accesses reasonably coalesced */
    const int sub = threadIdx.x & 7;
    const unsigned low = (1u << shift) - 1u;
    const uint64_t dropped = nbuckets << (shift + rbits);
    /* grid-stride over the rows: 8 work-items per row are more than one
     * launch holds (2^32) beyond 2^29 rows */
    const long long step = ((long long)gridDim.x * blockDim.x) >> 3;
    for (long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
         row < M; row += step) {
        const uint64_t tile = (uint64_t)(row / tile_rows);
        const unsigned rit = (unsigned)(row - (long long)tile * tile_rows);
        /* a long row's entries live beside the copy: dropped from the buckets */
        const bool out = nlong > 0 && in_sorted(long_row, nlong, (int)row);
        /* 64-bit entry index: k + 8 next to INT32_MAX (the entry count's limit) */
        for (int64_t k = (int64_t)irp[row] + sub, e = irp[row + 1]; k < e;
             k += 8) {
            const unsigned c = (unsigned)ja[k];
            key[k] = out ? dropped
                         : entry_key(bucket_id(tile, c >> shift, panels, pm_grid),
                                     c & low, rit, shift, rbits);
            val[k] = as[k];
        }
    }
}

/* HLL source, one lane per SLOT (a lane per row walks a hack block as wide as
 * its longest row: 32 lanes x 500 000 steps for a hub row, 0.14-0.33 s per
 * build).  Row of stored slot t: its block by bisection in off[], the row
 * inside the block from the layout */
__device__ __forceinline__ int hll_slot_row(int64_t t, int M, int nb, int col_major,
                                            const int64_t *__restrict__ off) {
    int lo = 0, hi = nb; /* largest b with off[b] <= t */
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= t)
            lo = mid;
        else
            hi = mid;
    }
    const int rows = min(32, M - lo * 32);
    const int64_t local = t - off[lo];
    const int w = hack_block_width(off, lo, rows);
    return lo * 32 + (col_major ? (int)(local % rows) : (int)(local / w));
}

/* One lane per STORED slot: does it hold an entry of the copy?  Only PAD
 * slots are left out (bitmap written when the pads were rewritten,
 * hll_kernels.hip) -- an explicit zero is an entry like any other, exactly
 * as from a CSR source -- and the slots of a long row, which is kept beside
 * the copy.  The kept slots are then COMPACTED before anything is sorted:
 * the format pads (slots / nnz = 8.6 on the KKT matrix, 10 on the power-law
 * rows), and 24 bytes of sort buffers per STORED slot were both the build's
 * time (the first multi-GB hipMalloc of a selector run: 1.3 s) and its
 * memory. */
__global__ void k_hll_keep_flags(int M, int nb, int64_t slots,
                                 const int *__restrict__ long_row, int nlong,
                                 int col_major, const int64_t *__restrict__ off,
                                 const unsigned *__restrict__ padmask,
                                 unsigned char *flag,
                                 unsigned long long *count) {
    /* grid-stride, one atomic per wavefront at the END (a count per 64 slots
     * would be 5M atomics on one address for the headline matrix: 50 ms) */
    unsigned long long mine = 0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < slots;
         t += (int64_t)gridDim.x * blockDim.x) {
        bool keep = !((padmask[t >> 5] >> (t & 31)) & 1u);
        if (keep && nlong > 0)
            keep = !in_sorted(long_row, nlong,
                              hll_slot_row(t, M, nb, col_major, off));
        flag[t] = keep ? 1 : 0;
        mine += keep;
    }
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1)
        mine += __shfl_down(mine, d, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0 && mine)
        atomicAdd(count, mine);
}

/* key and value of kept slot idx[r] (the list the compaction wrote,
 * ascending; identity: every stored slot is kept and no list exists) */
__global__ void k_keys_from_hll(int M, int nb, int64_t n, int tile_rows,
                                int panels, int shift, int rbits, int pm_grid,
                                int col_major, const int64_t *__restrict__ off,
                                const int *__restrict__ ja,
                                const double *__restrict__ as, int identity,
                                const unsigned *__restrict__ idx, uint64_t *key,
                                double *val) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n)
        return;
    const int64_t t = identity ? r : (int64_t)idx[r];
    const int row = hll_slot_row(t, M, nb, col_major, off);
    const uint64_t tile = (uint64_t)(row / tile_rows);
    const unsigned low = (1u << shift) - 1u;
    const unsigned c = (unsigned)ja[t];
    key[r] = entry_key(bucket_id(tile, c >> shift, panels, pm_grid), c & low,
                       (unsigned)(row - (int)tile * tile_rows), shift, rbits);
    val[r] = as[t];
}

/*
 * Slots of entry r of a bucket in ENT and VAL.  Buckets are cut into blocks
 * of 256 entries, one wavefront's chunk.  Lane L of the wavefront owns
 * entries L, 64 + L, 128 + L, 192 + L of the block: its u-th gather then
 * goes out together with those of entries 64u .. 64u + 63 -- CONSECUTIVE
 * entries of the column-sorted bucket, so lanes that want the same line of
 * x share one request (with 4 consecutive entries per lane instead, an
 * instruction spans 256 entries and shares nothing: 4x the L2 requests on a
 * band of 2^17 columns).  The block is stored so that the lane still gets
 * its four entries with one 16-byte load of ENT (slot 4L + u) and two of
 * VAL (2L + u for u < 2, 128 + 2L + u - 2 above): every load instruction of
 * the wavefront covers 1 KiB of whole cache lines, none of them twice (24
 * line requests per block instead of ~43 -- the CU's outstanding-miss slots
 * are what the kernels run out of).
 */
__device__ __forceinline__ int64_t block_ent_slot(int64_t r) {
    const int i = (int)(r & 255), u = i >> 6, L = i & 63;
    return (r & ~(int64_t)255) + 4 * L + u;
}

__device__ __forceinline__ int64_t block_val_slot(int64_t r) {
    const int i = (int)(r & 255), u = i >> 6, L = i & 63;
    return (r & ~(int64_t)255) + (u < 2 ? 2 * L + u : 128 + 2 * L + (u - 2));
}

/* sorted pair k -> its slots of the copy: ENT = row in tile << shift | column
 * in panel (both in the key), VAL = the value the sort carried.  Reads and
 * writes whole lines (a wavefront's 64 pairs land inside one 256-slot block) */
__global__ void k_place(int64_t n, int shift, int rbits,
                        const uint64_t *__restrict__ skey,
                        const double *__restrict__ sval,
                        const int64_t *__restrict__ raw,
                        const int64_t *__restrict__ bptr, unsigned *tent,
                        double *tval) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    const uint64_t kk = skey[k];
    const uint64_t bk = kk >> (shift + rbits); /* bucket */
    const int64_t rk = k - raw[bk];            /* rank inside the bucket */
    const unsigned col = (unsigned)(kk >> rbits) & ((1u << shift) - 1u);
    const unsigned rit = (unsigned)kk & ((1u << rbits) - 1u);
    tent[bptr[bk] + block_ent_slot(rk)] = (rit << shift) | col;
    tval[bptr[bk] + block_val_slot(rk)] = sval[k];
}

/* start of every (tile, panel) bucket: first position with bucket id >= b */
__global__ void k_bucket_bounds(int64_t buckets, int64_t n, int shift,
                                const uint64_t *__restrict__ key,
                                int64_t *bptr) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > buckets)
        return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((key[mid] >> shift) < (uint64_t)b)
            lo = mid + 1;
        else
            hi = mid;
    }
    bptr[b] = lo;
}

/* bucket sizes, and the sizes rounded up to `pad` slots (input of the prefix
 * sum that gives the padded starts); slot `buckets` closes the scan */
__global__ void k_bucket_sizes(int64_t buckets, int64_t pad,
                               const int64_t *__restrict__ raw, int *blen,
                               int64_t *padded) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > buckets)
        return;
    const int64_t len = b < buckets ? raw[b + 1] - raw[b] : 0;
    if (b < buckets)
        blen[b] = (int)len;
    padded[b] = (len + pad - 1) & ~(pad - 1);
}

/* first / last non-empty panel of every tile, and the widest such span */
__global__ void k_tile_span(int tiles, int panels,
                            const int *__restrict__ blen, int *first,
                            int *last, int *max_span) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles)
        return;
    int f = -1, l = -1;
    for (int p = 0; p < panels; ++p)
        if (blen[(int64_t)t * panels + p] > 0) {
            if (f < 0)
                f = p;
            l = p;
        }
    first[t] = f;
    last[t] = l;
    if (f >= 0)
        atomicMax(max_span, l - f + 1);
}

/*
 * Compact every tile's non-empty buckets; one thread per tile.  The list
 * order is the order in which the chain / steps schedules visit the buckets.
 *
 * K = panels: ascending panels.  K < panels (every tile's non-empty panels
 * lie within a span of K: a band): ascending (panel mod K) -- the RESIDUE
 * order.  Why: the tiles an XCD runs at one time are neighbours (grouped
 * order: 32 consecutive tiles), they start together and do equal work per
 * bucket.  Visiting each tile's own panels in ascending order puts tile t on
 * panel first(t) + q in phase q, and first(t) drifts along the diagonal: on
 * random W = 2^20 (tiles of 19552 rows, panels of 2^17 columns, 9-10 buckets
 * per tile) the 32 tiles of a group sit on 5-6 DIFFERENT 1 MiB panels of x
 * at any time, more than the XCD's 4 MiB L2 holds beside the entry stream,
 * and every line of x is fetched again by each of the 5-6 cohorts that pass
 * it one phase apart (x refetch 1.15 GB per SpMV, profiles/r01_m_w20.md;
 * that kernel runs at the fabric's ~6.6 TB/s of real traffic).  In residue
 * order phase r puts every tile on ITS panel congruent to r mod K; a group
 * spans fewer than 2K panels, so at most two distinct panels are live in an
 * XCD at a time and each panel is fetched once per group.
 */
__global__ void k_compact_buckets(int tiles, int panels, int K,
                                  const int64_t *__restrict__ bptr,
                                  const int *__restrict__ blen,
                                  const int *__restrict__ first,
                                  const int *__restrict__ last, int64_t *cb,
                                  int *cpanel, int *nbk) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles)
        return;
    int n = 0;
    const int f = first[t], l = last[t];
    if (f >= 0)
        for (int r = 0; r < K; ++r) {
            /* the panel of [f, f + K) congruent to r mod K */
            const int p = f + ((r - f) % K + K) % K;
            if (p > l)
                continue;
            int64_t b = bptr[(int64_t)t * panels + p];
            int64_t e = b + blen[(int64_t)t * panels + p];
            if (e > b) {
                cb[((int64_t)t * panels + n) * 2] = b;
                cb[((int64_t)t * panels + n) * 2 + 1] = e;
                cpanel[(int64_t)t * panels + n] = p;
                ++n;
            }
        }
    nbk[t] = n;
}

__global__ void k_max_int(int n, const int *__restrict__ v, int *out) {
    int m = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += gridDim.x * blockDim.x)
        m = max(m, v[i]);
    atomicMax(out, m);
}

/* process default of spmv_*_build_panels(): 0 steps, 1 sweep, 2 chain.  The
 * only process-wide setting of this file; builds that must not depend on it
 * pass spmv_panel_opts.sched (spmv_*_build_panels_opts / _as). */
static std::atomic<int> g_panel_sched{1};

extern "C" int spmv_set_panel_schedule(int sched) {
    if (sched < 0 || sched > 2)
        return -EINVAL;
    g_panel_sched.store(sched, std::memory_order_relaxed);
    return 0;
}

/* rows of y one workgroup can hold when `per_cu` of them share a CU's LDS */
static int sweep_tile_rows_max(int per_cu) {
    /* the workgroups of a CU share 160 KiB; each carries a 128-byte static
     * ring besides its tile */
    return (int)((160 * 1024 - 128 * per_cu) / per_cu / 8 / 32 * 32);
}

/* compute units of the current device; -ENODEV unless it is a gfx950 (the
 * XCD count and dispatch order the sweep schedule relies on, NUM_XCD, are
 * facts of that chip; the code objects are built for it alone anyway) */
static int device_cus(void) {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipGetDeviceProperties(&prop, dev) != hipSuccess)
        return -ENODEV;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return -ENODEV;
    return prop.multiProcessorCount;
}

/* tile height of the sweep schedule: the fewest rounds of (grid x tile_max)
 * rows that cover M, rows spread evenly over them */
static long long sweep_tile_rows(int M, int grid, int tile_max) {
    const long long rounds = ((long long)M + (long long)grid * tile_max - 1) /
                             ((long long)grid * tile_max);
    const long long wg = (rounds > 0 ? rounds : 1) * grid;
    long long tr = (((long long)M + wg - 1) / wg + 31) / 32 * 32;
    return tr < 32 ? 32 : tr;
}

/* Tile height of the chain / steps schedules with one workgroup per CU (tall
 * tiles): the fewest rounds of (CUs x max_rows) rows that cover M, rows
 * spread evenly, so that the last round is as full as the others.  10M rows:
 * 20448-row tiles are 489 tiles = 1.91 rounds of 256, 16384-row tiles 611 =
 * 2.39 rounds (the third round runs 20 % full: W = 2^20 0.883 ms); 19552
 * rows make exactly 512 tiles (0.774 ms), 13024 rows 768.  Returns max_rows
 * when the device cannot be queried. */
int panels_balanced_tile_rows(int M, int max_rows) {
    const int cus = device_cus();
    if (cus <= 0 || M <= 0)
        return max_rows;
    long long tr = sweep_tile_rows(M, cus, max_rows / 32 * 32);
    return (int)(tr > max_rows ? max_rows : tr);
}

/* start slot of every tile (and the end of the last one) */
__global__ void k_tile_starts(int tiles, int panels,
                              const int64_t *__restrict__ bptr, int64_t *out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t <= tiles)
        out[t] = bptr[(int64_t)t * panels];
}

/* contiguous tile ranges of about equal entry counts, one per XCD */
static int balance_xcd_ranges(spmv_panels *P) {
    int rc = 0;
    int64_t *d_st = NULL;
    const int tiles = P->tiles;
    std::vector<int64_t> st((size_t)tiles + 1, 0);
    if (tiles > 0) {
        HIP_TRY(hipMalloc((void **)&d_st, ((size_t)tiles + 1) * sizeof(int64_t)));
        hipLaunchKernelGGL(k_tile_starts, dim3((tiles + 256) / 256), dim3(256),
                           0, 0, tiles, P->panels, P->bptr, d_st);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(st.data(), d_st, ((size_t)tiles + 1) * sizeof(int64_t),
                          hipMemcpyDeviceToHost));
    }
    {
        const int64_t total = st[tiles] - st[0];
        P->xcd_first[0] = 0;
        for (int k = 1; k < NUM_XCD; ++k) {
            /* first tile whose start reaches k/8 of the entries; every tile
             * also counts one slot so that empty tiles spread evenly */
            const double want = (double)(total + tiles) * k / NUM_XCD;
            int lo = P->xcd_first[k - 1], hi = tiles;
            while (lo < hi) {
                const int mid = lo + (hi - lo) / 2;
                if ((double)(st[mid] - st[0] + mid) < want)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            P->xcd_first[k] = lo;
        }
        P->xcd_first[NUM_XCD] = tiles;
        P->xcd_max = 0;
        for (int k = 0; k < NUM_XCD; ++k)
            if (P->xcd_first[k + 1] - P->xcd_first[k] > P->xcd_max)
                P->xcd_max = P->xcd_first[k + 1] - P->xcd_first[k];
    }
fail:
    (void)hipFree(d_st);
    return rc;
}

static int bits_for(long long n) { /* smallest b with 2^b >= n */
    int b = 0;
    while ((1ll << b) < n)
        ++b;
    return b;
}

/* ------------------------------------------------------------------ */
/* long rows beside the copy (struct spmv_panels, "LONG ROWS")           */
/* ------------------------------------------------------------------ */
/* rows of more than `limit` entries -> list[1 + 2k] = row, [2 + 2k] = length;
 * list[0] counts them (may exceed `cap`: the host then gives up) */
__global__ void k_long_rows_csr(int M, int limit, int cap,
                                const int *__restrict__ irp, int *list) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= M)
        return;
    const int len = irp[row + 1] - irp[row];
    if (len > limit) {
        const int k = atomicAdd(list, 1);
        if (k < cap) {
            list[1 + 2 * k] = row;
            list[2 + 2 * k] = len;
        }
    }
}

/* HLL: only a hack block wider than `limit` can hold such a row; its real
 * length is the number of non-pad slots.  A WAVEFRONT per row: rows of
 * narrow blocks leave at once, the 32 rows of a wide block (a hub row: 10^5
 * slots and more) are counted by 64 lanes each */
__global__ void k_long_rows_hll(int M, int limit, int cap, int col_major,
                                const int64_t *__restrict__ off,
                                const unsigned *__restrict__ padmask, int *list) {
    const int lane = threadIdx.x & (WAVE - 1);
    /* grid-stride (wave-uniform): M wavefronts are more work-items than one
     * launch holds (2^32) once M reaches 2^26 */
    const long long step = (long long)gridDim.x * blockDim.x / WAVE;
    for (long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
         row < M; row += step) {
        const int b = (int)(row >> 5), i = (int)(row & 31);
        const int rows = min(32, M - b * 32);
        const int64_t o = off[b];
        const int w = hack_block_width(off, b, rows);
        if (w <= limit)
            continue;
        int len = 0;
        for (int j = lane; j < w; j += WAVE) {
            const int64_t t =
                o + (col_major ? (int64_t)j * rows + i : (int64_t)i * w + j);
            len += !((padmask[t >> 5] >> (t & 31)) & 1u);
        }
#pragma unroll
        for (int d = WAVE / 2; d > 0; d >>= 1)
            len += __shfl_down(len, d, WAVE);
        if (lane == 0 && len > limit) {
            const int k = atomicAdd(list, 1);
            if (k < cap) {
                list[1 + 2 * k] = (int)row;
                list[2 + 2 * k] = len;
            }
        }
    }
}

/* one workgroup per long row copies its entries, in row order */
__global__ void k_long_copy_csr(const int *__restrict__ long_row,
                                const int *__restrict__ long_ptr,
                                const int *__restrict__ irp,
                                const int *__restrict__ ja,
                                const double *__restrict__ as, int *lja,
                                double *las) {
    const int h = blockIdx.x, row = long_row[h];
    const int src = irp[row], dst = long_ptr[h], len = long_ptr[h + 1] - dst;
    for (int k = threadIdx.x; k < len; k += blockDim.x) {
        lja[dst + k] = ja[src + k];
        las[dst + k] = as[src + k];
    }
}

/* ... HLL: one WAVEFRONT per long row walks its slots 64 at a time and
 * compacts the non-pad ones (ballot + prefix count), keeping their order */
__global__ void k_long_copy_hll(int M, int col_major,
                                const int *__restrict__ long_row,
                                const int *__restrict__ long_ptr,
                                const int64_t *__restrict__ off,
                                const int *__restrict__ ja,
                                const double *__restrict__ as,
                                const unsigned *__restrict__ padmask, int *lja,
                                double *las) {
    const int h = blockIdx.x, row = long_row[h], lane = threadIdx.x;
    const int b = row >> 5, i = row & 31;
    const int rows = min(32, M - b * 32);
    const int64_t o = off[b];
    const int w = hack_block_width(off, b, rows);
    int dst = long_ptr[h];
    for (int j0 = 0; j0 < w; j0 += WAVE) {
        const int j = j0 + lane;
        bool keep = false;
        int64_t t = 0;
        if (j < w) {
            t = o + (col_major ? (int64_t)j * rows + i : (int64_t)i * w + j);
            keep = !((padmask[t >> 5] >> (t & 31)) & 1u);
        }
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int at = dst + __popcll(m & ((1ull << lane) - 1ull));
            lja[at] = ja[t];
            las[at] = as[t];
        }
        dst += __popcll(m);
    }
}

/* the second launch: workgroup g sums segment g (entries [beg, end) of long
 * row seg[g].x); the last segment of a row to arrive adds the row's partial
 * sums in a fixed order and OVERWRITES y[row] (the tile kernels wrote 0) */
__global__ void __launch_bounds__(256)
    k_long_rows(const int2 *__restrict__ seg, int nseg,
                const int *__restrict__ long_row,
                const int *__restrict__ long_ptr,
                const int *__restrict__ seg0, const int *__restrict__ lja,
                const double *__restrict__ las, const double *__restrict__ x,
                double *__restrict__ y, double *part, unsigned long long *cnt,
                unsigned epoch) {
    __shared__ double wsum[4];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int h = seg[g].x, beg = seg[g].y;
    const int row_end = long_ptr[h + 1];
    const int end = min(beg + PANELS_LONG_SEG, row_end);
    double acc = strided_dot<256, 4>(lja, las, x, beg, end, tid);
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1)
        acc += __shfl_down(acc, d, WAVE);
    if ((tid & (WAVE - 1)) == 0)
        wsum[tid / WAVE] = acc;
    __syncthreads();
    __shared__ int s_last;
    const int g0 = seg0[h];
    if (tid == 0) {
        const double t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        const int n =
            (row_end - long_ptr[h] + PANELS_LONG_SEG - 1) / PANELS_LONG_SEG;
        __hip_atomic_store(part + g, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = epoch_arrive(cnt + h, epoch) == (unsigned)n ? n : 0;
    }
    __syncthreads();
    if (s_last && tid < WAVE) { /* one wavefront adds the row's partial sums,
                                   in a fixed order (hip_common.h) */
        const double sum = wave_ordered_sum(part + g0, s_last, tid);
        if (tid == 0) {
            y[long_row[h]] = sum;
            epoch_rearm(cnt + h, epoch);
        }
    }
}

/* find the long rows of the source and copy them beside the (future) copy;
 * P->nlong stays 0 when there are none (or absurdly many: then they stay in
 * the buckets).  Called before the keys are made. */
static int long_rows_extract(spmv_panels *P, int M, int nb,
                             const int *irp_or_null, const int64_t *off_or_null,
                             int col_major, const int *ja, const double *as,
                             const unsigned *padmask) {
    int rc = 0;
    int *d_list = NULL;
    std::vector<int> list((size_t)2 * PANELS_LONG_MAX + 1, 0);
    std::vector<std::pair<int, int>> rows; /* (row, length) */
    std::vector<int> h_row, h_ptr, h_seg0;
    std::vector<int2> h_seg;
    (void)nb;
    if (M <= 0)
        return 0;
    HIP_TRY(hipMalloc((void **)&d_list, list.size() * sizeof(int)));
    HIP_TRY(hipMemset(d_list, 0, sizeof(int)));
    if (irp_or_null)
        hipLaunchKernelGGL(k_long_rows_csr, dim3((M + 255) / 256), dim3(256), 0,
                           0, M, PANELS_LONG_ROW, PANELS_LONG_MAX, irp_or_null,
                           d_list);
    else
        hipLaunchKernelGGL(k_long_rows_hll,
                           dim3((unsigned)std::min<long long>(
                               ((long long)M * WAVE + 255) / 256, 1 << 22)),
                           dim3(256), 0, 0, M, PANELS_LONG_ROW, PANELS_LONG_MAX,
                           col_major, off_or_null, padmask, d_list);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(list.data(), d_list, list.size() * sizeof(int),
                      hipMemcpyDeviceToHost));
    if (list[0] <= 0 || list[0] > PANELS_LONG_MAX)
        goto fail; /* none / too many: nothing beside the copy */
    for (int k = 0; k < list[0]; ++k)
        rows.push_back({list[1 + 2 * k], list[2 + 2 * k]});
    std::sort(rows.begin(), rows.end());
    {
        int64_t at = 0;
        for (size_t h = 0; h < rows.size(); ++h) {
            h_row.push_back(rows[h].first);
            h_ptr.push_back((int)at);
            h_seg0.push_back((int)h_seg.size());
            for (int b = 0; b < rows[h].second; b += PANELS_LONG_SEG)
                h_seg.push_back(make_int2((int)h, (int)at + b));
            at += rows[h].second;
            if (at > (int64_t)INT32_MAX)
                goto fail; /* keep them inside */
        }
        h_ptr.push_back((int)at);
        P->long_nnz = at;
    }
    P->nlong = (int)rows.size();
    P->nlong_seg = (int)h_seg.size();
    HIP_TRY(hipMalloc((void **)&P->long_row, h_row.size() * sizeof(int)));
    HIP_TRY(hipMalloc((void **)&P->long_ptr, h_ptr.size() * sizeof(int)));
    HIP_TRY(hipMalloc((void **)&P->long_seg0, h_seg0.size() * sizeof(int)));
    HIP_TRY(hipMalloc((void **)&P->long_seg, h_seg.size() * sizeof(int2)));
    HIP_TRY(hipMalloc((void **)&P->long_ja, (size_t)P->long_nnz * sizeof(int)));
    HIP_TRY(hipMalloc((void **)&P->long_as, (size_t)P->long_nnz * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&P->long_part, h_seg.size() * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&P->long_cnt,
                      h_row.size() * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(P->long_part, 0, h_seg.size() * sizeof(double)));
    HIP_TRY(hipMemset(P->long_cnt, 0,
                      h_row.size() * sizeof(unsigned long long)));
    HIP_TRY(hipMemcpy(P->long_row, h_row.data(), h_row.size() * sizeof(int),
                      hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(P->long_ptr, h_ptr.data(), h_ptr.size() * sizeof(int),
                      hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(P->long_seg0, h_seg0.data(), h_seg0.size() * sizeof(int),
                      hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(P->long_seg, h_seg.data(), h_seg.size() * sizeof(int2),
                      hipMemcpyHostToDevice));
    if (irp_or_null)
        hipLaunchKernelGGL(k_long_copy_csr, dim3(P->nlong), dim3(256), 0, 0,
                           P->long_row, P->long_ptr, irp_or_null, ja, as,
                           P->long_ja, P->long_as);
    else
        hipLaunchKernelGGL(k_long_copy_hll, dim3(P->nlong), dim3(WAVE), 0, 0, M,
                           col_major, P->long_row, P->long_ptr, off_or_null, ja,
                           as, padmask, P->long_ja, P->long_as);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
fail:
    (void)hipFree(d_list);
    if (rc) {
        P->nlong = P->nlong_seg = 0;
        P->long_nnz = 0;
    }
    return rc;
}

/*
 * Block pool of one selector run.  spmv_*_autotune builds four or five
 * candidates of ONE matrix; each build needs ~11 GB of sort temporaries and
 * 3.8 GB for the copy at config-3 size, and the losers are freed again.  With
 * plain hipMalloc / hipFree every build returns its memory to the driver and
 * takes fresh memory back -- usually in milliseconds, but one allocation in
 * ten or so took 1.3-1.6 s (round 3's unexplained 4.2-4.6 s selector runs;
 * round 4's phase log pins it on the allocations: "alloc+keys 1.536 s" /
 * "gather 1.329 s" = the hipMalloc of the copy, `profiles/r04_selector_phases.txt`).
 * Inside panels_pool_begin() .. panels_pool_end() the big blocks of builds and
 * of freed candidates are parked and handed to the next build instead (a
 * block is reused for a request of 89-100 % of its size); outside a selector
 * run nothing is pooled.
 */
struct build_pool {
    struct blk {
        void *p;
        size_t bytes;
    };
    std::vector<blk> parked; /* free, reusable */
    std::vector<blk> out;    /* handed out (size remembered for parking) */
};
static thread_local build_pool *g_pool = NULL;

void panels_pool_begin(void) {
    if (!g_pool)
        g_pool = new build_pool();
}

void panels_pool_end(void) {
    if (!g_pool)
        return;
    for (size_t i = 0; i < g_pool->parked.size(); ++i)
        (void)hipFree(g_pool->parked[i].p);
    delete g_pool; /* blocks still out belong to the kept copy: plain memory */
    g_pool = NULL;
}

static hipError_t big_malloc(void **p, size_t bytes) {
    if (g_pool) {
        for (size_t i = 0; i < g_pool->parked.size(); ++i) {
            const size_t have = g_pool->parked[i].bytes;
            if (have >= bytes && have - bytes <= have / 9) {
                *p = g_pool->parked[i].p;
                g_pool->out.push_back(g_pool->parked[i]);
                g_pool->parked.erase(g_pool->parked.begin() + (long)i);
                return hipSuccess;
            }
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess && g_pool && !g_pool->parked.empty()) {
        /* out of memory with blocks parked: give them back and retry */
        for (size_t i = 0; i < g_pool->parked.size(); ++i)
            (void)hipFree(g_pool->parked[i].p);
        g_pool->parked.clear();
        (void)hipGetLastError();
        e = hipMalloc(p, bytes);
    }
    if (e == hipSuccess && g_pool)
        g_pool->out.push_back({*p, bytes});
    return e;
}

static void big_free(void *p) {
    if (!p)
        return;
    if (g_pool)
        for (size_t i = 0; i < g_pool->out.size(); ++i)
            if (g_pool->out[i].p == p) {
                g_pool->parked.push_back(g_pool->out[i]);
                g_pool->out.erase(g_pool->out.begin() + (long)i);
                return;
            }
    (void)hipFree(p);
}

/* host-clock seconds of the last panels_build on this thread, by phase
 * (allocations + keys, radix sort, bucket tables, gather into the copy):
 * what spmv_*_autotune appends to its log */
static thread_local char g_build_phases[160];
const char *panels_last_build_phases(void) { return g_build_phases; }
static double build_now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static int panels_build(int M, int N, int64_t slots, const spmv_panel_opts *o,
                        int nb, const int *irp_or_null,
                        const int64_t *off_or_null, int col_major,
                        const int *ja, const double *as,
                        const unsigned *padmask, spmv_panels **out) {
    int rc = 0;
    *out = NULL;
    if (slots > (int64_t)INT32_MAX)
        return -EOVERFLOW;
    spmv_panel_opts dflt;
    spmv_panel_opts_default(&dflt);
    if (!o)
        o = &dflt;
    if (o->struct_size != (int)sizeof(spmv_panel_opts))
        return -EINVAL; /* caller built against another header (ABI note) */
    if (o->sched > 2 || o->panel_cols < 0 || o->tile_rows < 0 ||
        o->sweep_wgs_per_cu < 0 || o->sweep_wgs_per_cu > 8 ||
        o->reserve_cus < 0 || o->lds_min < 0 || o->lds_min > BIG_LDS_BYTES ||
        o->tile_order < 0 || o->tile_order > 2 || o->sweep_layout < -1 ||
        o->sweep_layout > 1 || o->bucket_order < 0 || o->bucket_order > 1 ||
        o->deterministic < 0 || o->deterministic > 2)
        return -EINVAL;
    const int panel_cols = o->panel_cols, tile_rows = o->tile_rows;
    int sched = o->sched;
    if (sched < 0)
        sched = g_panel_sched.load(std::memory_order_relaxed);
    const int sweep = sched == 1;
    /* tile height.  steps: the taller the tile, the more entries of a bucket
     * share a line of x (fewer L2 requests) but the fewer workgroups there
     * are and the more LDS each holds; best measured per matrix: 2048 rows
     * for a band of 2^14 columns (0.68 ms), 8192 for 2^17 (0.84), 16384 for
     * 2^20 (1.18) and for skewed rows (0.21) -- spmv_*_autotune tries these.
     * sweep: sweep_tile_rows(). */
    long long tr = TILE_ROWS_STEPS;
    int grid = 0, tile_max = 20448 /* steps: 160 KiB of LDS */, per_cu = 0;
    int want_shift = 18; /* 2^18 columns = 2 MiB of x: half of an XCD's L2 */
    if (panel_cols > 0)
        want_shift = bits_for((long long)panel_cols + 1) - 1; /* floor(log2) */
    if (sweep) {
        /* one 1024-lane workgroup per CU with a 160 KiB tile, or two of up
         * to 512 lanes with 80 KiB tiles.  The tall tile halves the panel
         * (row and column index share 32 bits) but puts twice the entries
         * on a line of x, which the column-sorted buckets turn into fewer
         * L2 requests: 1.61 vs 1.77 ms on config 3.  It needs buckets that
         * fill its 4096-slot chunks, so it is taken only above 4000 entries
         * per bucket (80 M columns, 1071 per bucket: 4.85 vs 3.35 ms). */
        int cus = device_cus();
        if (cus < 0)
            return cus;
        if (o->reserve_cus > 0) /* keep at least one XCD's worth of CUs */
            cus = cus - o->reserve_cus >= NUM_XCD ? cus - o->reserve_cus : NUM_XCD;
        per_cu = o->sweep_wgs_per_cu;
        if (per_cu == 0) {
            const int tm = sweep_tile_rows_max(1);
            const long long tr1 = sweep_tile_rows(M, cus, tm);
            int sh = want_shift;
            if (sh > 32 - bits_for(tr1))
                sh = 32 - bits_for(tr1);
            const double tiles1 = (double)(((long long)M + tr1 - 1) / tr1);
            const double panels1 =
                (double)((((int64_t)(N > 0 ? N : 1) - 1) >> sh) + 1);
            per_cu = (double)slots / (tiles1 * panels1) >= 4000.0
                         ? 1 : SWEEP_WG_PER_CU;
        }
        grid = cus * per_cu;
        tile_max = sweep_tile_rows_max(per_cu);
        tr = sweep_tile_rows(M, grid, tile_max);
#ifdef SPMV_ABLATIONS
        /* round-6 probe (tools/tall_tile_probe.py): a sweep copy with tiles
         * TALLER than a CU's LDS can hold, launched with variant ablation 5 / 6
         * (LDS index aliased, y WRONG by design): the L2 request pattern of
         * accumulators that do not exist yet */
        if (const char *e = getenv("SPMV_ABL_SWEEP_TILE_ROWS")) {
            const long long v = atoll(e) / 32 * 32;
            if (v >= 32 && v < (1 << 20)) {
                per_cu = 1;
                grid = cus;
                tr = v;
            }
        }
#endif
    }
    if (!sweep && tile_rows >= 32 && tile_rows <= tile_max)
        tr = tile_rows / 32 * 32;
    const int rbits = bits_for(tr);
    int shift = want_shift;
    if (shift > 32 - rbits) /* row-in-tile and column-in-panel share a word */
        shift = 32 - rbits;
    const int panels = (int)((((int64_t)(N > 0 ? N : 1) - 1) >> shift) + 1);
    const int tiles = (int)(((long long)M + tr - 1) / tr);
    /* bucket tables are indexed with int (hipCUB scan length) */
    if ((uint64_t)(tiles > 0 ? tiles : 1) * (uint64_t)panels >=
        (uint64_t)INT32_MAX)
        return -EOVERFLOW;

    spmv_panels *P = (spmv_panels *)calloc(1, sizeof *P);
    if (!P)
        return -ENOMEM;
    P->N = N;
    P->shift = shift;
    P->panels = panels;
    P->tile_rows = (int)tr;
    P->tiles = tiles;
    P->sweep = sweep;
    P->chain = sched == 2;
    P->grid = sweep ? (grid < tiles ? grid : (tiles > 0 ? tiles : 1)) : 0;
    P->wgs_per_cu = per_cu;
    P->reserve_cus = sweep ? o->reserve_cus : 0;
    /* panel-major is the default (-1): 1.448 vs 1.464 ms on config 3, 2.88
     * vs 2.96 ms on one shard of the 80M-column problem, and the masked tail
     * blocks re-read a cached line instead of fetching another bucket */
    P->pmajor = sweep && o->sweep_layout != 0;
    P->lds_min = o->lds_min;
    P->bucket_order = o->bucket_order;
    /* 0 = default: ordered additions on sweep layouts (free there: the
     * hand-offs hide under the request-bound loop, +-2 %), arrival order on
     * chain / steps (+3..+10 % there); 1 = always, 2 = never */
    P->det = o->deterministic == 1 || (o->deterministic == 0 && sweep);
    P->order = sweep ? 0 : o->tile_order;
    /* bucket ids: tile-major, or panel-major inside rounds of P->grid tiles */
    const int pm_grid = P->pmajor ? P->grid : 0;
    const int64_t nbuckets =
        pm_grid ? (int64_t)((tiles + pm_grid - 1) / pm_grid) * panels * pm_grid
                : (int64_t)tiles * panels;
    if ((uint64_t)nbuckets >= (uint64_t)INT32_MAX) {
        free(P);
        return -EOVERFLOW;
    }
    uint64_t *key[2] = {NULL, NULL};
    double *sv[2] = {NULL, NULL}; /* the values the sort carries */
    unsigned *idx = NULL;         /* HLL source with pads: kept slots */
    uint64_t *skey = NULL;
    double *sval = NULL;
    void *tmp = NULL;
    size_t tmp_bytes = 0;
    int64_t *raw = NULL, *padded = NULL; /* unpadded bucket starts, scan input */
    int *span = NULL;                    /* per-tile first / last panel */
    int64_t n_sort = slots; /* sorted pairs: every CSR entry / every KEPT HLL slot */
    unsigned char *flag = NULL;
    unsigned long long *d_count = NULL;
    int64_t total = 0;
    double tp[5] = {build_now_s(), 0, 0, 0, 0};
    g_build_phases[0] = 0;

    rc = long_rows_extract(P, M, nb, irp_or_null, off_or_null, col_major, ja,
                           as, padmask);
    if (rc)
        goto fail;
    if (!irp_or_null && slots > 0) {
        /* HLL source: flag the slots that hold entries, count them, compact
         * their indices (ascending) into idx[0] -- see k_hll_keep_flags */
        unsigned long long h_count = 0;
        HIP_TRY(big_malloc((void **)&flag, (size_t)slots));
        HIP_TRY(hipMalloc((void **)&d_count, sizeof *d_count));
        HIP_TRY(hipMemset(d_count, 0, sizeof *d_count));
        hipLaunchKernelGGL(k_hll_keep_flags,
                           dim3((unsigned)std::min<int64_t>((slots + 255) / 256,
                                                            8192)),
                           dim3(256), 0,
                           0, M, nb, slots, P->long_row, P->nlong, col_major,
                           off_or_null, padmask, flag, d_count);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(&h_count, d_count, sizeof h_count,
                          hipMemcpyDeviceToHost));
        n_sort = (int64_t)h_count;
    }
    {
        const size_t na = (size_t)(n_sort > 0 ? n_sort : 1);
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(big_malloc((void **)&key[k], na * sizeof(uint64_t)));
            HIP_TRY(big_malloc((void **)&sv[k], na * sizeof(double)));
        }
        if (!irp_or_null && n_sort != slots)
            HIP_TRY(big_malloc((void **)&idx, na * sizeof(unsigned)));
    }
    skey = key[0];
    sval = sv[0];
    if (slots > 0) {
        if (irp_or_null) {
            hipLaunchKernelGGL(k_keys_from_csr,
                               dim3((unsigned)std::min<long long>(
                                   ((long long)M * 8 + 255) / 256, 1 << 23)),
                               dim3(256), 0, 0, M, (int)tr, panels, shift,
                               rbits, pm_grid, (uint64_t)nbuckets, P->long_row,
                               P->nlong, irp_or_null, ja, as, key[0], sv[0]);
        } else if (n_sort > 0) {
            const int identity = n_sort == slots; /* the format padded nothing */
            if (!identity) {
                void *st = NULL;
                size_t stb = 0;
                hipcub::CountingInputIterator<unsigned> every_slot(0u);
                /* the count is known: d_count is only written again */
                HIP_TRY(hipcub::DeviceSelect::Flagged(st, stb, every_slot, flag,
                                                      idx, d_count,
                                                      (int)slots, 0));
                HIP_TRY(big_malloc(&st, stb ? stb : 16));
                hipError_t e1 = hipcub::DeviceSelect::Flagged(
                    st, stb, every_slot, flag, idx, d_count, (int)slots, 0);
                if (e1 == hipSuccess)
                    e1 = hipDeviceSynchronize();
                big_free(st);
                HIP_TRY(e1);
            }
            big_free(flag); /* 1 byte per stored slot: not needed any more */
            flag = NULL;
            hipLaunchKernelGGL(k_keys_from_hll,
                               dim3((unsigned)((n_sort + 255) / 256)), dim3(256),
                               0, 0, M, nb, n_sort, (int)tr, panels, shift,
                               rbits, pm_grid, col_major, off_or_null, ja, as,
                               identity, idx, key[0], sv[0]);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        tp[1] = build_now_s();
        {
            big_free(idx); /* the keys are built: the list is not needed */
            idx = NULL;
            hipcub::DoubleBuffer<uint64_t> dk(key[0], key[1]);
            hipcub::DoubleBuffer<double> dv(sv[0], sv[1]);
            /* sort only the bits in use: bucket ids up to tiles*panels (the
             * id of dropped slots) above `shift` column bits; the row field
             * below them rides along unsorted (entry_key) */
            const int kb = shift + rbits;
            int end_bit = kb + 1;
            while (end_bit < 64 && (((uint64_t)nbuckets) >> (end_bit - kb)))
                ++end_bit;
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(
                NULL, tmp_bytes, dk, dv, (int)n_sort, rbits, end_bit, 0));
            HIP_TRY(big_malloc(&tmp, tmp_bytes ? tmp_bytes : 16));
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(
                tmp, tmp_bytes, dk, dv, (int)n_sort, rbits, end_bit, 0));
            HIP_TRY(hipDeviceSynchronize());
            skey = dk.Current();
            sval = dv.Current();
        }
    }
    tp[2] = build_now_s();
    {
        const int64_t buckets = nbuckets;
        const unsigned gb = (unsigned)((buckets + 256) / 256);
        HIP_TRY(hipMalloc((void **)&raw, ((size_t)buckets + 1) * sizeof(int64_t)));
        HIP_TRY(hipMalloc((void **)&padded,
                          ((size_t)buckets + 1) * sizeof(int64_t)));
        HIP_TRY(hipMalloc((void **)&P->bptr,
                          ((size_t)buckets + 1) * sizeof(int64_t)));
        HIP_TRY(hipMalloc((void **)&P->blen, ((size_t)buckets + 1) * sizeof(int)));
        hipLaunchKernelGGL(k_bucket_bounds, dim3(gb), dim3(256), 0, 0, buckets,
                           n_sort, shift + rbits, skey, raw);
        hipLaunchKernelGGL(k_bucket_sizes, dim3(gb), dim3(256), 0, 0, buckets,
                           (int64_t)256, raw, P->blen, padded);
        HIP_TRY(hipGetLastError());
        {
            void *t2 = NULL;
            size_t t2b = 0;
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(t2, t2b, padded, P->bptr,
                                                     (int)(buckets + 1), 0));
            HIP_TRY(hipMalloc(&t2, t2b ? t2b : 16));
            hipError_t e2 = hipcub::DeviceScan::ExclusiveSum(
                t2, t2b, padded, P->bptr, (int)(buckets + 1), 0);
            (void)hipDeviceSynchronize();
            (void)hipFree(t2);
            HIP_TRY(e2);
        }
        /* compacted per-tile bucket lists of the chain / steps schedules.
         * A sweep copy has none (its kernel reads bptr / blen only): cb,
         * cpanel and nbk stay NULL and panels_launch / the setters refuse to
         * run such a copy in another schedule. */
        if (!sweep) {
            HIP_TRY(hipMalloc((void **)&P->cb,
                              ((size_t)buckets + 1) * 2 * sizeof(int64_t)));
            HIP_TRY(hipMalloc((void **)&P->cpanel,
                              ((size_t)buckets + 1) * sizeof(int)));
            HIP_TRY(hipMalloc((void **)&P->nbk,
                              ((size_t)tiles + 1) * sizeof(int)));
            HIP_TRY(hipMemset(P->nbk, 0, ((size_t)tiles + 1) * sizeof(int)));
            if (tiles > 0) {
                /* span[0] first, [tiles] last, [2 tiles] widest span */
                HIP_TRY(hipMalloc((void **)&span,
                                  (2 * (size_t)tiles + 1) * sizeof(int)));
                HIP_TRY(hipMemset(span + 2 * (size_t)tiles, 0, sizeof(int)));
                hipLaunchKernelGGL(k_tile_span, dim3((tiles + 255) / 256),
                                   dim3(256), 0, 0, tiles, panels, P->blen,
                                   span, span + tiles, span + 2 * (size_t)tiles);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpy(&P->span, span + 2 * (size_t)tiles,
                                  sizeof(int), hipMemcpyDeviceToHost));
                /* residue order whenever the tiles do not touch every panel
                 * (see k_compact_buckets; with span == panels it would be the
                 * identity); bucket_order 1 keeps ascending panels */
                P->residue = o->bucket_order != 1 && P->span > 1 &&
                             P->span < panels;
                hipLaunchKernelGGL(k_compact_buckets, dim3((tiles + 255) / 256),
                                   dim3(256), 0, 0, tiles, panels,
                                   P->residue ? P->span : panels, P->bptr,
                                   P->blen, span, span + tiles, P->cb,
                                   P->cpanel, P->nbk);
                hipLaunchKernelGGL(k_max_int, dim3(64), dim3(256), 0, 0, tiles,
                                   P->nbk, P->nbk + tiles);
                HIP_TRY(hipGetLastError());
            }
            HIP_TRY(hipMemcpy(&P->max_nbk, P->nbk + tiles, sizeof(int),
                              hipMemcpyDeviceToHost));
        }
        /* dropped slots sort behind the last bucket */
        HIP_TRY(hipMemcpy(&P->nnz, raw + buckets, sizeof(int64_t),
                          hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&total, P->bptr + buckets, sizeof total,
                          hipMemcpyDeviceToHost));
    }
    /* The kernels index the copy's slots with UNSIGNED 32 bits (fk / fe /
     * blk, `fk + CH >= fe`): room for every matrix the API can describe
     * (INT32_MAX entries) plus its bucket padding, as long as `slot + chunk`
     * cannot wrap -- 3e9 leaves 1.29e9 of head room for a chunk of 16 Ki. */
    if (total + SWEEP_TAIL > (int64_t)3000000000LL) {
        rc = -EOVERFLOW;
        goto fail;
    }
    P->total = total;
    tp[3] = build_now_s();
    {
        const size_t m = (size_t)total + SWEEP_TAIL; /* zeros behind the data */
        HIP_TRY(big_malloc((void **)&P->ent, m * sizeof(unsigned)));
        HIP_TRY(big_malloc((void **)&P->val, m * sizeof(double)));
        HIP_TRY(hipMemset(P->ent, 0, m * sizeof(unsigned)));
        HIP_TRY(hipMemset(P->val, 0, m * sizeof(double)));
    }
    if (P->nnz > 0) {
        const unsigned g = (unsigned)((P->nnz + 255) / 256);
        hipLaunchKernelGGL(k_place, dim3(g), dim3(256), 0, 0, P->nnz, shift,
                           rbits, skey, sval, raw, P->bptr, P->ent, P->val);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
    }
    if (!sweep) {
        rc = balance_xcd_ranges(P);
        if (rc)
            goto fail;
    }
    if (sweep) {
        const size_t rounds = ((size_t)tiles + P->grid - 1) / P->grid;
        P->phase_cnt_bytes = NUM_XCD * (rounds ? rounds : 1) * (size_t)panels *
                             CNT_STRIDE * sizeof(int);
        HIP_TRY(hipMalloc((void **)&P->phase_cnt, P->phase_cnt_bytes));
    }
    tp[4] = build_now_s();
    snprintf(g_build_phases, sizeof g_build_phases,
             "%lld of %lld slots sorted: alloc+keys %.3f s, sort %.3f, bucket "
             "tables %.3f, place %.3f",
             (long long)n_sort, (long long)slots, tp[1] > 0 ? tp[1] - tp[0] : 0.0,
             tp[1] > 0 ? tp[2] - tp[1] : 0.0, tp[3] - tp[2], tp[4] - tp[3]);
    *out = P;
    P = NULL;
fail:
    for (int k = 0; k < 2; ++k) {
        big_free(key[k]);
        big_free(sv[k]);
    }
    big_free(idx);
    big_free(tmp);
    big_free(flag);
    (void)hipFree(d_count);
    (void)hipFree(raw);
    (void)hipFree(padded);
    (void)hipFree(span);
    panels_free(P);
    return rc;
}

/* ------------------------------------------------------------------ */
/* schedule "sweep": one persistent launch                               */
/* ------------------------------------------------------------------ */
__device__ __forceinline__ void phase_arrive(int *cnt, int n = 1) {
    __hip_atomic_fetch_add(cnt, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/* bounded: a miss only costs L2 locality */
/* returns false when the bound expired: the grid is not co-resident (another
 * kernel holds CUs) -- the caller stops waiting for the rest of the launch */
__device__ __forceinline__ bool phase_wait(const int *cnt, int want, int spin) {
    for (int i = 0; i < spin; ++i) {
        if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >=
            want)
            return true;
        __builtin_amdgcn_s_sleep(8);
    }
    return false;
}

/*
 * One arrival per WORKGROUP and phase: the wavefront that leaves panel q
 * last arrives for all of them.  (One arrival per wavefront was tried and
 * measured slower: 256 same-address atomics per phase and XCD serialise in
 * one L2 channel; the selector fell back to the chain schedule.)  The
 * workgroup-local tally is a ring of 32 slots indexed by phase.  While a
 * wavefront still waits on phase counters it cannot run more than lag + 3
 * <= 10 phases ahead of the slowest wavefront of its XCD -- its own
 * workgroup's included -- so a slot is never reused before it was reset (the
 * reset is an atomic exchange by the last arriver).  A wavefront whose wait
 * expired runs on unsynchronised and could lap the ring; the tally may then
 * miscount, which can only make other waits expire early (bounded, costs L2
 * locality, never correctness).
 */
#define WAVE_RING 32
__device__ __forceinline__ void wg_arrive(int *ring, int q, int waves,
                                          int *cnt_q) {
    const int old = atomicAdd(&ring[q & (WAVE_RING - 1)], 1);
    if (old == waves - 1) {
        atomicExch(&ring[q & (WAVE_RING - 1)], 0);
        phase_arrive(cnt_q);
    }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

/*
 * A lane owns Q groups of 4 consecutive entries per chunk: one 16-byte load
 * of ENT and two of VAL per group (layout: block_ent_slot).  Three
 * chunk buffers rotate: while chunk c is gathered and added, the loads of
 * c+1 are a whole iteration old and those of c+2 are issued behind c's
 * gathers -- the CU's texture path serves requests in order, so a stream
 * load queues behind every gather issued before it and needs that lead.
 * Chunks follow each other across panel boundaries.
 *
 * Every vector-memory instruction of the loop body is UNCONDITIONAL (lanes
 * past the end of a bucket read the slots that follow -- the arrays carry
 * SWEEP_TAIL slots of slack -- and are masked afterwards; the phase poll is
 * issued for every chunk and only looked at after a panel's last one): with
 * loads under divergent or uniform branches the compiler can no longer
 * count the in-order vmcnt queue and waits for everything, which
 * serialises gathers and stream (measured 2.2 ms vs the sum of both).
 * The poll for the NEXT panel's entry condition thus returns with the
 * gathers and costs no drain; only a miss falls into the (bounded) spin.
 *
 * Tried in round 2 and dropped on measurement (gpurun r2c6, same run): a
 * wavefront-granular form -- each wavefront walks 256-slot blocks of the
 * tile's entry list straight across bucket boundaries, all bookkeeping in
 * SGPRs -- which removes the masked tail chunk of every bucket (the third
 * chunk of a ~8500-entry bucket is 7 % full).  Config 3, W = N: 1.57 vs
 * 1.56 ms at 512 lanes, 1.49 vs 1.63 at 256 (noise-level gain); one shard of
 * the 80M-column problem (610 panels, ~1000-entry buckets): 5.42 vs 3.01 ms
 * -- the per-panel scalar work of every wavefront costs more than the tail
 * chunks did.  Also dropped: one phase arrival per WAVEFRONT instead of per
 * workgroup (256 same-address atomics per phase and XCD; the selector fell
 * back to the chain schedule).
 */

/*
 * Deterministic mode (spmv_panel_opts.deterministic): the additions of one
 * chunk into the tile's LDS slice happen one wavefront after the other, in
 * wavefront order, chunk after chunk -- a turn counter in LDS that a
 * wavefront waits for, adds under, and passes on.  A wavefront's own LDS
 * instructions execute in order, and the lanes of one ds_add_f64 that meet on
 * an address are served in a fixed order, so every row's products are added
 * in ONE order per copy: the same bits on every launch.  No workgroup
 * barrier: while a wavefront waits for its turn the others go on gathering
 * and loading; all wavefronts of a workgroup consume the same number of
 * chunks (the chunk walk is workgroup-uniform), so the turn always arrives.
 */
/* SPMV_DET_SLEEP: s_sleep argument between two polls of the turn counter
 * (x 64 clocks); 0 = poll back to back.  Experiment knob of round 6
 * (`make abl EXTRA=-DSPMV_DET_SLEEP=0`, tools/det_cost.py): see EXPERIMENTS */
#ifndef SPMV_DET_SLEEP
#define SPMV_DET_SLEEP 1
#endif
__device__ __forceinline__ void det_wait(int *turn, int want) {
    while (__hip_atomic_load(turn, __ATOMIC_ACQUIRE,
                             __HIP_MEMORY_SCOPE_WORKGROUP) != want) {
#if SPMV_DET_SLEEP > 0
        __builtin_amdgcn_s_sleep(SPMV_DET_SLEEP);
#endif
    }
}

/* Ordered mode: a wavefront takes its turn only with its products IN HAND --
 * the gathers of x it is waiting for must not be waited for inside the turn,
 * where the seven other wavefronts of the workgroup queue behind it.  The
 * empty asm makes every product a value the compiler has to have computed
 * (hence its s_waitcnt for the gather passed) before the turn counter is
 * polled; the stream loads of the chunk after next, issued later, stay in
 * flight (the vector-memory queue returns in order). */
__device__ __forceinline__ void det_have(double v) {
    asm volatile("" ::"v"(v));
}

__device__ __forceinline__ void det_pass(int *turn, int next) {
    __hip_atomic_store(turn, next, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int Q> struct sweep_chunk {
    u32x4 en[Q];
    f64x2 va[Q], vb[Q];
    int live[Q];      /* entry u of the lane exists iff live > 64 u */
    int p;            /* wave-uniform: phase / panel (>= panels: past the end) */
    int pa;           /* wave-uniform, sweep: the panel visited in phase p */
    bool first, last; /* wave-uniform: first / last chunk of its bucket */
};

template <int NT, int Q, int ABL = 0, bool DET = false>
/* ABL: timing ablations, 1 = no LDS add, 2 = gathers from one 8 KiB window
 * (L1 hits), 8 = tile taller than its LDS slice: row index aliased into 16384
 * rows (y wrong by design; round-6 tall-tile probe); DET: deterministic mode */
__global__ void __launch_bounds__(NT)
    k_tiles_sweep(int M, int tile_rows, int tiles, int panels, int shift,
                  int lag, int spin, int stagger, int pmajor, unsigned total,
                  const int64_t *__restrict__ bptr,
                  const int *__restrict__ blen,
                  const unsigned *__restrict__ tent,
                  const double *__restrict__ tval,
                  const double *__restrict__ x, double *__restrict__ y,
                  int *phase_cnt) {
    extern __shared__ double ytile[];
    __shared__ int wave_done[WAVE_RING];
    constexpr int WAVES = NT / WAVE;
    constexpr unsigned CH = NT * Q * 4;
    const int tid = threadIdx.x;
    const int grid = gridDim.x;
    /* workgroups are dealt to the XCDs round-robin; a phase is complete when
     * every workgroup of the XCD has arrived (wg_arrive) */
    const int xcd = blockIdx.x % NUM_XCD;
    const int n_x = grid / NUM_XCD + (xcd < (grid % NUM_XCD) ? 1 : 0);
    const int rounds = (tiles + grid - 1) / grid;
    int *cnt = phase_cnt + (size_t)xcd * rounds * panels * CNT_STRIDE;
    const unsigned lowmask = (ABL & 2) ? 1023u : (1u << shift) - 1u;
    __shared__ int det_turn;
    int det_seq = tid / WAVE; /* my next turn: chunk index * WAVES + wavefront */
    if (tid < WAVE_RING)
        wave_done[tid] = 0;
    if (tid == 0)
        det_turn = 0;
    bool synced = true; /* false once a wait expired: run on unsynchronised
                           (costs L2 locality) instead of paying the bound at
                           every panel */

    for (int r = 0; r < rounds; ++r) {
        const int t = r * grid + blockIdx.x;
        const int q0 = r * panels;
        if (t >= tiles) { /* no tile this round: arrive at all its phases */
            if (tid == 0)
                for (int p = 0; p < panels; ++p)
                    phase_arrive(cnt + (size_t)(q0 + p) * CNT_STRIDE);
            continue;
        }
        for (int i = tid; i < ((ABL & 8) && tile_rows > 16384 ? 16384 : tile_rows);
             i += NT)
            ytile[i] = 0.0;
        __syncthreads();

        /* bucket (this tile, panel p): tile-major tables, or panel-major
         * inside the round (layout 1: every workgroup's bucket of a panel
         * lies in one compact region) */
        const int64_t bbase = pmajor ? (int64_t)q0 * grid + blockIdx.x
                                     : (int64_t)t * panels;
        const int64_t bstep = pmajor ? grid : 1;
        auto bp = [&](int p) { return bptr[bbase + (int64_t)p * bstep]; };
        auto bl = [&](int p) { return blen[bbase + (int64_t)p * bstep]; };
        bool ready = !(lag > 0 && q0 >= lag); /* panel 0 of a later round */

        /* position of the next chunk to load (wave-uniform).  All XCDs visit
         * the panels in the same order: the first one to touch a panel pulls
         * it into the Infinity Cache for the other seven.  (`stagger`, tuning
         * bit 12: phase p of XCD k visits panel p + k * panels / 8 instead --
         * no gain at 10 M columns, 2.97 -> 3.79 ms at 80 M.) */
        const int poff = stagger ? (int)((long long)xcd * panels / NUM_XCD) : 0;
        int fp = 0, fpa = poff % panels;
        unsigned fk = (unsigned)bp(fpa), fe = fk + (unsigned)bl(fpa);
        unsigned fb = fk; /* start of the bucket being read */
        bool ffirst = true;

        auto fill = [&](sweep_chunk<Q> &c) {
            c.p = fp;
            c.pa = fpa;
            c.first = ffirst;
            c.last = fk + CH >= fe;
#pragma unroll
            for (int g = 0; g < Q; ++g) {
                /* the wavefront's block of 256 slots: ENT in entry order,
                 * VAL permuted (block_val_slot) so that each of the three
                 * loads reads 1 KiB of whole lines */
                unsigned blk = fk + ((unsigned)g * NT + (tid & ~(WAVE - 1))) * 4u;
                const unsigned lane = tid & (WAVE - 1);
                c.live[g] = (int)(fe - blk) - (int)lane; /* > 64u: entry u */
                /* layout 1: a block past the bucket's end would fetch
                 * ANOTHER tile's bucket (pure waste; tile-major it prefetches
                 * this tile's next bucket): re-read the bucket's first block */
                if (pmajor && blk >= fe && fe > fb)
                    blk = fb;
                if (ABL & 4) /* stream from a 192 KiB window: L2 hits */
                    blk &= 0x3FFFu;
                c.en[g] = __builtin_nontemporal_load(
                    (const u32x4 *)(tent + blk + lane * 4u));
                c.va[g] = __builtin_nontemporal_load(
                    (const f64x2 *)(tval + blk + lane * 2u));
                c.vb[g] = __builtin_nontemporal_load(
                    (const f64x2 *)(tval + blk + 128u + lane * 2u));
            }
            /* advance */
            if (fp < panels) {
                if (!c.last) {
                    fk += CH;
                    ffirst = false;
                } else {
                    fp += 1;
                    fpa = fpa + 1 < panels ? fpa + 1 : 0;
                    ffirst = true;
                    if (fp < panels) {
                        fk = (unsigned)bp(fpa);
                        fe = fk + (unsigned)bl(fpa);
                    } else { /* past the end: zeros of the tail */
                        fk = total;
                        fe = total;
                    }
                    fb = fk;
                }
            }
        };

        /* gather and add chunk `c`, loading the chunk after next into `f` */
        auto step = [&](sweep_chunk<Q> &c, sweep_chunk<Q> &f) {
            const int q = q0 + c.p;
            if (c.first && !ready && synced)
                synced = phase_wait(cnt + (size_t)(q - lag) * CNT_STRIDE, n_x,
                                    spin);
            const double *xp = x + ((int64_t)c.pa << shift);
            double pr[Q][4], w[Q][4];
            unsigned rr[Q][4];
            int on[Q];
#pragma unroll
            for (int g = 0; g < Q; ++g) {
                on[g] = c.live[g];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned col = 64 * u < on[g] ? (c.en[g][u] & lowmask) : 0u;
                    pr[g][u] = xp[col];
                    rr[g][u] = c.en[g][u] >> shift;
                    if (ABL & 8)
                        rr[g][u] &= 16383u;
                }
                w[g][0] = c.va[g][0];
                w[g][1] = c.va[g][1];
                w[g][2] = c.vb[g][0];
                w[g][3] = c.vb[g][1];
            }
            const bool last = c.last;
            /* entry condition of the next panel; looked at only if `last` */
            const int qn = (last && lag > 0 && q + 1 >= lag) ? q + 1 - lag : 0;
            const int polled = __hip_atomic_load(cnt + (size_t)qn * CNT_STRIDE,
                                                 __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
            fill(f);
            /* (sweep: the turn is taken BEFORE the gathers are waited for --
             * with products in hand first, as the chain / steps kernels do,
             * this request-bound loop measured 1.47 instead of 1.43-1.45 ms:
             * here the wait for the turn hides under the gathers' latency) */
            if (DET)
                det_wait(&det_turn, det_seq);
#pragma unroll
            for (int g = 0; g < Q; ++g)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double prod = pr[g][u] * w[g][u];
                    if (ABL & 1) {
                        if (64 * u < on[g] && prod == 1.2345e300)
                            ytile[rr[g][u]] = 1.0;
                    } else if (64 * u < on[g]) {
                        unsafeAtomicAdd(&ytile[rr[g][u]], prod);
                    }
                }
            if (DET) {
                det_pass(&det_turn, det_seq + 1);
                det_seq += WAVES;
            }
            if (last) {
                ready = lag <= 0 || q + 1 < lag || polled >= n_x;
                if ((tid & (WAVE - 1)) == 0)
                    wg_arrive(wave_done, q, WAVES, cnt + (size_t)q * CNT_STRIDE);
            }
        };

        sweep_chunk<Q> A, B, C;
        fill(A);
        fill(B);
        for (;;) {
            step(A, C);
            if (B.p >= panels)
                break;
            step(B, A);
            if (C.p >= panels)
                break;
            step(C, B);
            if (A.p >= panels)
                break;
        }
        __syncthreads();
        const int64_t row0 = (int64_t)t * tile_rows;
        for (int i = tid; i < tile_rows && row0 + i < M; i += NT)
            __builtin_nontemporal_store(ytile[(ABL & 8) ? (i & 16383) : i],
                                        y + row0 + i);
        __syncthreads();
    }
}

/* ------------------------------------------------------------------ */
/* schedule "steps": one launch per step                                 */
/* ------------------------------------------------------------------ */
/*
 * In step s workgroup t adds the s-th non-empty bucket of tile t into the
 * tile's slice of y (through LDS; a tile has one owner per launch and the
 * launches are stream-ordered).  Step 0 starts every slice from zero -- also
 * for tiles without entries -- so y needs no memset and is only re-read by
 * tiles that reach a second panel.  Same entry layout and load discipline as
 * the sweep kernel: blocks of 256 slots per wavefront, 16-byte loads over
 * whole lines, every vector load unconditional, the loads of chunk c+1 behind
 * the gathers of chunk c.
 */
template <int NT, int Q, bool DET = false>
__global__ void __launch_bounds__(NT)
    k_tiles_step(int M, int tile_rows, int panels, int shift, int step,
                 int tiles_hw, xcd_ranges xr,
                 const int64_t *__restrict__ cb, const int *__restrict__ cpanel,
                 const int *__restrict__ nbk, const unsigned *__restrict__ tent,
                 const double *__restrict__ tval, const double *__restrict__ x,
                 double *__restrict__ y) {
    extern __shared__ double ytile[];
    __shared__ int det_turn;
    constexpr unsigned CH = NT * Q * 4;
    constexpr int WAVES = NT / WAVE;
    const int tid = threadIdx.x;
    int det_seq = tid / WAVE;
    if (tid == 0)
        det_turn = 0; /* published by the barrier behind the slice's load */
    /* XCD-contiguous tile ranges of equal work (xcd_ranges): the tiles an
     * XCD runs at one time are neighbours, so their step-th panels coincide
     * or are adjacent */
    /* tiles_hw > 0: hardware order (tile = workgroup index); < 0: groups of
     * G = -tiles_hw >> 24 consecutive tiles per XCD, the groups dealt to the
     * XCDs round-robin (tiles = -tiles_hw & 0xffffff): neighbouring tiles
     * share an L2 AND the chip as a whole advances through one region */
    int t, t_end;
    if (tiles_hw > 0) {
        t = (int)blockIdx.x;
        t_end = tiles_hw;
    } else if (tiles_hw < 0) {
        const int G = (-tiles_hw) >> 24, k = blockIdx.x / NUM_XCD;
        t = ((k / G) * NUM_XCD + (int)(blockIdx.x % NUM_XCD)) * G + k % G;
        t_end = (-tiles_hw) & 0xffffff;
    } else {
        t = xr.first[blockIdx.x % NUM_XCD] + (int)(blockIdx.x / NUM_XCD);
        t_end = xr.first[blockIdx.x % NUM_XCD + 1];
    }
    if (t >= t_end)
        return; /* beyond the tiles / this XCD's range */
    const int64_t row0 = (int64_t)t * tile_rows;
    if (step >= nbk[t]) {
        if (step == 0) /* a tile without entries: its rows are zero */
            for (int i = tid; i < tile_rows && row0 + i < M; i += NT)
                y[row0 + i] = 0.0;
        return;
    }
    const unsigned b = (unsigned)cb[((int64_t)t * panels + step) * 2];
    const unsigned e = (unsigned)cb[((int64_t)t * panels + step) * 2 + 1];
    const double *xp = x + ((int64_t)cpanel[(int64_t)t * panels + step] << shift);
    const unsigned lowmask = (1u << shift) - 1u;
    const unsigned lane = tid & (WAVE - 1);
    const unsigned wbase = (tid & ~(WAVE - 1)) * 4u;

    auto fill = [&](sweep_chunk<Q> &c, unsigned k0) {
#pragma unroll
        for (int g = 0; g < Q; ++g) {
            const unsigned blk = k0 + (unsigned)g * NT * 4u + wbase;
            c.live[g] = (int)(e - blk) - (int)lane; /* > 64u: entry u */
            c.en[g] = __builtin_nontemporal_load(
                (const u32x4 *)(tent + blk + lane * 4u));
            c.va[g] = __builtin_nontemporal_load(
                (const f64x2 *)(tval + blk + lane * 2u));
            c.vb[g] = __builtin_nontemporal_load(
                (const f64x2 *)(tval + blk + 128u + lane * 2u));
        }
    };
    auto consume = [&](sweep_chunk<Q> &c, sweep_chunk<Q> &f, unsigned knext) {
        double pr[Q][4], w[Q][4];
        unsigned rr[Q][4];
        int on[Q];
#pragma unroll
        for (int g = 0; g < Q; ++g) {
            on[g] = c.live[g];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned col = 64 * u < on[g] ? (c.en[g][u] & lowmask) : 0u;
                pr[g][u] = xp[col];
                rr[g][u] = c.en[g][u] >> shift;
            }
            w[g][0] = c.va[g][0];
            w[g][1] = c.va[g][1];
            w[g][2] = c.vb[g][0];
            w[g][3] = c.vb[g][1];
        }
        fill(f, knext); /* past the bucket: slack slots, masked by `live` */
        if (DET) {
#pragma unroll
            for (int g = 0; g < Q; ++g)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    pr[g][u] *= w[g][u];
                    w[g][u] = 1.0;
                    det_have(pr[g][u]);
                }
            det_wait(&det_turn, det_seq);
        }
#pragma unroll
        for (int g = 0; g < Q; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (64 * u < on[g])
                    unsafeAtomicAdd(&ytile[rr[g][u]],
                                    DET ? pr[g][u] : pr[g][u] * w[g][u]);
        if (DET) {
            det_pass(&det_turn, det_seq + 1);
            det_seq += WAVES;
        }
    };

    sweep_chunk<Q> A, B;
    fill(A, b);
    /* the first entries and the y slice are fetched together; the x gathers
     * follow the barrier (issued before it they delay the slice, whose loads
     * return in order behind them) */
    if (step == 0)
        for (int i = tid; i < tile_rows; i += NT)
            ytile[i] = 0.0;
    else
        for (int i = tid; i < tile_rows; i += NT)
            ytile[i] = row0 + i < M ? y[row0 + i] : 0.0;
    __syncthreads();
    for (unsigned k0 = b;;) {
        consume(A, B, k0 + CH);
        k0 += CH;
        if (k0 >= e)
            break;
        consume(B, A, k0 + CH);
        k0 += CH;
        if (k0 >= e)
            break;
    }
    __syncthreads();
    for (int i = tid; i < tile_rows && row0 + i < M; i += NT)
        y[row0 + i] = ytile[i];
}

/* ------------------------------------------------------------------ */
/* schedule "chain": the steps layout in ONE launch                      */
/* ------------------------------------------------------------------ */
/*
 * Workgroup t walks ALL non-empty buckets of tile t in panel order with the
 * tile's slice of y in LDS and writes it once: no read-modify-write of y, no
 * launch boundaries, loads running ahead across bucket boundaries.  There is
 * no phase control: neighbouring tiles (XCD-contiguous ranges) start
 * together and do the same amount of work per bucket, so on a banded /
 * clustered matrix, where a tile has a handful of buckets, they stay on
 * neighbouring panels by themselves.  On a matrix without locality (dozens
 * of buckets per tile) they drift apart -- that is what the sweep schedule's
 * phase counters are for.
 */
template <int NT, int Q, bool DET = false>
__global__ void __launch_bounds__(NT)
    k_tiles_chain(int M, int tile_rows, int panels, int shift, unsigned total,
                  int tiles_hw, xcd_ranges xr,
                  const int64_t *__restrict__ cb, const int *__restrict__ cpanel,
                  const int *__restrict__ nbk, const unsigned *__restrict__ tent,
                  const double *__restrict__ tval, const double *__restrict__ x,
                  double *__restrict__ y) {
    extern __shared__ double ytile[];
    __shared__ int det_turn;
    constexpr unsigned CH = NT * Q * 4;
    constexpr int WAVES = NT / WAVE;
    const int tid = threadIdx.x;
    int det_seq = tid / WAVE;
    if (tid == 0)
        det_turn = 0; /* published by the barrier behind the slice's zeroing */
    /* tiles_hw > 0: hardware order (tile = workgroup index); < 0: groups of
     * G = -tiles_hw >> 24 consecutive tiles per XCD, the groups dealt to the
     * XCDs round-robin (tiles = -tiles_hw & 0xffffff): neighbouring tiles
     * share an L2 AND the chip as a whole advances through one region */
    int t, t_end;
    if (tiles_hw > 0) {
        t = (int)blockIdx.x;
        t_end = tiles_hw;
    } else if (tiles_hw < 0) {
        const int G = (-tiles_hw) >> 24, k = blockIdx.x / NUM_XCD;
        t = ((k / G) * NUM_XCD + (int)(blockIdx.x % NUM_XCD)) * G + k % G;
        t_end = (-tiles_hw) & 0xffffff;
    } else {
        t = xr.first[blockIdx.x % NUM_XCD] + (int)(blockIdx.x / NUM_XCD);
        t_end = xr.first[blockIdx.x % NUM_XCD + 1];
    }
    if (t >= t_end)
        return; /* beyond the tiles / this XCD's range */
    const int64_t row0 = (int64_t)t * tile_rows;
    const int nb = nbk[t];
    const int64_t *tcb = cb + (int64_t)t * panels * 2;
    const int *tpan = cpanel + (int64_t)t * panels;
    const unsigned lowmask = (1u << shift) - 1u;
    const unsigned lane = tid & (WAVE - 1);
    const unsigned wbase = (tid & ~(WAVE - 1)) * 4u;

    /* position of the next chunk to load (wave-uniform) */
    int fs = 0, fpan = nb > 0 ? tpan[0] : 0;
    unsigned fk = nb > 0 ? (unsigned)tcb[0] : total;
    unsigned fe = nb > 0 ? (unsigned)tcb[1] : total;

    auto fill = [&](sweep_chunk<Q> &c) {
        c.p = fs < nb ? fpan : -1;
#pragma unroll
        for (int g = 0; g < Q; ++g) {
            const unsigned blk = fk + (unsigned)g * NT * 4u + wbase;
            c.live[g] = (int)(fe - blk) - (int)lane;
            c.en[g] = __builtin_nontemporal_load(
                (const u32x4 *)(tent + blk + lane * 4u));
            c.va[g] = __builtin_nontemporal_load(
                (const f64x2 *)(tval + blk + lane * 2u));
            c.vb[g] = __builtin_nontemporal_load(
                (const f64x2 *)(tval + blk + 128u + lane * 2u));
        }
        if (fs < nb) {
            fk += CH;
            if (fk >= fe) {
                fs += 1;
                if (fs < nb) {
                    fk = (unsigned)tcb[2 * fs];
                    fe = (unsigned)tcb[2 * fs + 1];
                    fpan = tpan[fs];
                } else { /* past the end: zeros of the tail */
                    fk = fe = total;
                }
            }
        }
    };
    auto consume = [&](sweep_chunk<Q> &c, sweep_chunk<Q> &f) {
        const double *xp = x + ((int64_t)c.p << shift);
        double pr[Q][4], w[Q][4];
        unsigned rr[Q][4];
        int on[Q];
#pragma unroll
        for (int g = 0; g < Q; ++g) {
            on[g] = c.live[g];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned col = 64 * u < on[g] ? (c.en[g][u] & lowmask) : 0u;
                pr[g][u] = xp[col];
                rr[g][u] = c.en[g][u] >> shift;
            }
            w[g][0] = c.va[g][0];
            w[g][1] = c.va[g][1];
            w[g][2] = c.vb[g][0];
            w[g][3] = c.vb[g][1];
        }
        fill(f);
        if (DET) {
#pragma unroll
            for (int g = 0; g < Q; ++g)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    pr[g][u] *= w[g][u];
                    w[g][u] = 1.0;
                    det_have(pr[g][u]);
                }
            det_wait(&det_turn, det_seq);
        }
#pragma unroll
        for (int g = 0; g < Q; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (64 * u < on[g])
                    unsafeAtomicAdd(&ytile[rr[g][u]],
                                    DET ? pr[g][u] : pr[g][u] * w[g][u]);
        if (DET) {
            det_pass(&det_turn, det_seq + 1);
            det_seq += WAVES;
        }
    };

    sweep_chunk<Q> A, B;
    fill(A);
    for (int i = tid; i < tile_rows; i += NT)
        ytile[i] = 0.0;
    __syncthreads();
    for (;;) {
        if (A.p < 0)
            break;
        consume(A, B);
        if (B.p < 0)
            break;
        consume(B, A);
    }
    __syncthreads();
    for (int i = tid; i < tile_rows && row0 + i < M; i += NT)
        y[row0 + i] = ytile[i];
}

/* tiles above 64 KiB of LDS need the opt-in, once per kernel and device */
template <auto Kernel> static int allow_big_lds(void) {
    /* one bit per device; setting the attribute twice from two host threads
     * that race here is harmless (idempotent), losing a set bit is not */
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    HIP_RET(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        /* dynamic + static LDS must stay within 160 KiB: the sweep kernels
         * carry a 128-byte static ring (asking for 160 KiB - 64 made the
         * call fail with hipErrorInvalidValue once the ring grew from 32 to
         * 128 bytes).  BIG_LDS_BYTES = the tallest tile, 20448 rows. */
        HIP_RET(hipFuncSetAttribute(reinterpret_cast<const void *>(Kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    BIG_LDS_BYTES));
        done.fetch_or(bit, std::memory_order_release);
    }
    return 0;
}

static int panels_launch_tiles(const spmv_panels *P, int M, int waves,
                               int variant, const double *x, double *y,
                               hipStream_t s);

int panels_launch(const spmv_panels *P, int M, int waves, int variant,
                  const double *x, double *y, hipStream_t s) {
    const int rc = panels_launch_tiles(P, M, waves, variant, x, y, s);
    if (rc || !P->nlong)
        return rc;
    /* the rows kept beside the copy, on the same stream: the tile kernels
     * have written 0 for them, this launch overwrites */
    hipLaunchKernelGGL(k_long_rows, dim3(P->nlong_seg), dim3(256), 0, s,
                       P->long_seg, P->nlong_seg, P->long_row, P->long_ptr,
                       P->long_seg0, P->long_ja, P->long_as, x, y, P->long_part,
                       P->long_cnt, next_launch_epoch(&P->launch_epoch));
    return hip_errno(hipGetLastError());
}

static int panels_launch_tiles(const spmv_panels *P, int M, int waves,
                               int variant, const double *x, double *y,
                               hipStream_t s) {
    if (!P)
        return -EINVAL;
    if (M == 0)
        return 0;
#ifndef SPMV_ABLATIONS
    /* product build: bit 0 flips chain <-> steps, bits 1 / 2 force a tile
     * order (spmv_engine.h).  Everything else this function understands --
     * lag override (4-6), no phase wait (7), the ABL arms whose result is
     * WRONG by design (8-10), group counts (11), staggered panels (12),
     * group sizes (14-15) -- exists only in a -DSPMV_ABLATIONS build
     * (`make abl`; tools/sweep.py, tools/pmc.sh load that flavour) */
    if (variant & ~(1 | 2 | 4 | SPMV_VARIANT_TIMING_BITS))
        return -EINVAL;
#endif
    (void)hipGetLastError(); /* an earlier caller's unread error is not ours */
    size_t lds = (size_t)P->tile_rows * sizeof(double);
#ifdef SPMV_ABLATIONS
    if (P->sweep && lds > (size_t)BIG_LDS_BYTES) /* tall-tile probe: aliased */
        lds = 16384 * sizeof(double);
#endif
    if (waves <= 0)
        waves = P->waves_hint;
    if ((size_t)P->lds_min > lds) /* tuning: caps workgroups per CU */
        lds = (size_t)P->lds_min;
    if (P->sweep) {
        /* variant (tuning): bits 4-6 lag override (1..7), bit 7 no phase
         * wait, bits 8-10 ablations, bit 11 the other group count, bit 12
         * staggered panel order */
        int lag = (variant >> 4) & 7;
        if (lag == 0) /* measured best: 6 for 1 MiB panels, 3 for 2 MiB ones,
                         7 when there are hundreds of them (80 M columns:
                         3.30 -> 2.98 ms) */
            lag = P->wgs_per_cu == 1 ? 6 : P->panels >= 256 ? 7 : 3;
        if (variant & 128)
            lag = 0;
        HIP_RET(hipMemsetAsync(P->phase_cnt, 0, P->phase_cnt_bytes, s));
#define SW_(NTHR, QQ, A, D)                                                    \
    do {                                                                       \
        if (int rc_ = allow_big_lds<&k_tiles_sweep<NTHR, QQ, A, D>>())         \
            return rc_;                                                        \
        hipLaunchKernelGGL((k_tiles_sweep<NTHR, QQ, A, D>), dim3(P->grid),    \
                           dim3(NTHR), lds, s, M, P->tile_rows, P->tiles,      \
                           P->panels, P->shift, lag, SWEEP_SPIN_MAX,           \
                           !!(variant & 4096), P->pmajor, (unsigned)P->total,  \
                           P->bptr, P->blen, P->ent, P->val, x, y,             \
                           P->phase_cnt);                                      \
    } while (0)
#define SW(NTHR, QQ, A)                                                        \
    do {                                                                       \
        if (P->det && (A) == 0)                                                \
            SW_(NTHR, QQ, 0, true);                                            \
        else                                                                   \
            SW_(NTHR, QQ, A, false);                                           \
    } while (0)
        const int two = !((variant >> 11) & 1); /* 2 groups of 4 per lane */
#ifdef SPMV_ABLATIONS /* timing ablations (results WRONG by design for 2, 3,
                         7): compiled only by `make abl` */
        const int abl = (variant >> 8) & 7;
        if (abl == 1) { SW(256, 1, 1); }
        else if (abl == 2) { SW(256, 1, 2); }
        else if (abl == 3) { SW(256, 1, 3); }
        else if (abl == 4) { SW(256, 1, 4); }
        else if (abl == 7) { SW(256, 1, 7); }
        else if (abl == 5) { /* tall-tile probe, the production launch shapes */
            if (waves > 8) SW(1024, 2, 8);
            else if (waves > 0 && waves < 8) SW(256, 2, 8);
            else SW(512, 2, 8);
        }
        else if (abl == 6) {
            if (waves > 8) SW(1024, 1, 8);
            else if (waves > 0 && waves < 8) SW(256, 1, 8);
            else SW(512, 1, 8);
        }
        else
#endif
        if (P->wgs_per_cu == 1) {
            /* 512 lanes x 2 groups measured best with the 160 KiB tile
             * (1.53 ms on config 3; 1024 x 1: 1.60); bit 11 flips the groups */
            if (waves > 8) { if (variant & 2048) SW(1024, 2, 0); else SW(1024, 1, 0); }
            else if (waves > 0 && waves < 8) { if (variant & 2048) SW(256, 1, 0); else SW(256, 2, 0); }
            else { if (variant & 2048) SW(512, 1, 0); else SW(512, 2, 0); }
        }
        else if (waves > 0 && waves < 8) { if (two) SW(256, 2, 0); else SW(256, 1, 0); }
        else if (waves >= 8) { if (two) SW(512, 2, 0); else SW(512, 1, 0); }
        else {
            /* default: the chunk (threads x groups x 4 slots) that wastes
             * few lanes on the average bucket; 512 x 2 measured best on
             * config 3 (8000 entries per bucket) */
            const double per_bucket =
                (double)P->nnz / ((double)P->tiles * (double)P->panels);
            if (per_bucket >= 6000.0) SW(512, 2, 0);
            else if (per_bucket >= 3000.0) SW(512, 1, 0);
            else SW(256, 1, 0);
        }
#undef SW
#undef SW_
        return hip_errno(hipGetLastError());
    }
    if (P->tiles <= 0 || P->xcd_max <= 0)
        return 0;
    if (!P->cb || !P->cpanel || !P->nbk)
        return -EINVAL; /* not a chain / steps copy */
    xcd_ranges xr;
    memcpy(xr.first, P->xcd_first, sizeof xr.first);
    /* Which tile a workgroup runs (workgroups are dealt to the XCDs
     * round-robin).  0 GROUPED (default): groups of 32 consecutive tiles --
     * one per CU of an XCD -- per XCD, the groups dealt round-robin, so
     * neighbouring tiles share an L2 AND the eight XCDs together advance
     * through one region of the matrix (and through rows of any density
     * together: no XCD idles on a matrix that is denser in one half).
     * 1 HARDWARE order: tile = workgroup index.  2 XCD-CONTIGUOUS ranges of
     * equal work.  Measured (round 2, tile 8192 unless noted; ms):
     *                      grouped  hardware  contiguous
     *   banded 10M x 32     0.615     0.619     0.642
     *   random W = 2^11     0.646     0.642     0.665
     *   random W = 2^17     0.672     0.755     0.654
     *   random W = 2^20     0.925     1.686     0.922   (20448 rows: 0.79 / 1.12 / 0.78)
     *   27-point stencil    0.473     0.485     0.499
     *   skewed rows 8.3M    0.171     0.176     0.174
     * variant bit 1 forces hardware order, bit 2 the contiguous ranges,
     * bits 14-15 a group size of 32 / 64 / 16 (experiments). */
    const int gsel = (variant >> 14) & 3;
    int ord = gsel ? 0 : (variant & 2) ? 1 : (variant & 4) ? 2 : P->order;
    if (ord == 0 && P->tiles >= (1 << 24))
        ord = 2; /* the packed (group, tiles) argument holds 24 bits of tiles */
    const int G = gsel == 2 ? 64 : gsel == 3 ? 16 : 32; /* < 128: 7 bits */
    const int order_arg = ord == 0 ? -((G << 24) | P->tiles)
                          : ord == 1 ? P->tiles : 0;
    const unsigned order_grid =
        ord == 0 ? (unsigned)((P->tiles + NUM_XCD * G - 1) / (NUM_XCD * G)) *
                       NUM_XCD * G
        : ord == 1 ? (unsigned)P->tiles : (unsigned)(NUM_XCD * P->xcd_max);
    if (P->chain != !!(variant & 1)) { /* variant bit 0 flips the stored mode */
        const double per_bucket_c =
            (double)P->nnz / ((double)P->tiles *
                              (double)(P->max_nbk > 0 ? P->max_nbk : 1));
#define CHN_(NTHR, QQ, D)                                                      \
    do {                                                                       \
        if (int rc_ = allow_big_lds<&k_tiles_chain<NTHR, QQ, D>>())            \
            return rc_;                                                        \
        hipLaunchKernelGGL((k_tiles_chain<NTHR, QQ, D>),                      \
                           dim3(order_grid),                                   \
                           dim3(NTHR), lds, s, M, P->tile_rows, P->panels,     \
                           P->shift, (unsigned)P->total, order_arg,            \
                           xr, P->cb, P->cpanel, P->nbk, P->ent, P->val, x,    \
                           y);                                                 \
    } while (0)
/* deterministic: several groups of 4 entries per lane and turn -- the
 * hand-offs of the turn counter are what the mode costs, and they go with
 * the number of turns -- at most 512 lanes.  Four groups when the tile fills
 * a CU's LDS by itself (one workgroup per CU: W = 2^20, 19552-row tiles,
 * 0.797 ms vs 0.900 with two groups and 0.728 in the default mode), two when
 * two or more workgroups share the CU and hide each other's hand-offs (four
 * groups cost them occupancy: 8192-row tiles at W = 2^17 0.690 vs 0.836 ms;
 * default mode 0.604).  profiles/r05_det_cost.md */
#define CHN(NTHR, QQ)                                                          \
    do {                                                                       \
        if (P->det && (size_t)P->tile_rows * sizeof(double) > 80 * 1024) {     \
            if ((NTHR) <= 256)                                                 \
                CHN_(256, 4, true);                                            \
            else                                                               \
                CHN_(512, 4, true);                                            \
        } else if (P->det) {                                                   \
            if ((NTHR) <= 256)                                                 \
                CHN_(256, 2, true);                                            \
            else                                                               \
                CHN_(512, 2, true);                                            \
        } else                                                                 \
            CHN_(NTHR, QQ, false);                                             \
    } while (0)
        if (variant & 2048) { /* tuning: two groups of 4 per lane */
            if (waves > 0 && waves < 8) CHN(256, 2);
            else CHN(512, 2);
        }
        else if (waves > 0 && waves < 8) CHN(256, 1);
        else if (waves > 8) CHN(1024, 1);
        else if (waves == 8 || per_bucket_c >= 3000.0) CHN(512, 1);
        else CHN(256, 1);
#undef CHN
#undef CHN_
        return hip_errno(hipGetLastError());
    }
    /* launch `step` handles the step-th NON-EMPTY bucket of every tile: a
     * matrix whose rows reach over k panels needs k launches, all tiles busy
     * in each of them; step 0 also zeroes the rows of empty tiles */
    const int steps = P->max_nbk > 0 ? P->max_nbk : 1;
    const double per_bucket =
        (double)P->nnz / ((double)P->tiles * (double)steps);
    for (int p = 0; p < steps; ++p) {
#define ST_(NTHR, QQ, D)                                                       \
    do {                                                                       \
        if (int rc_ = allow_big_lds<&k_tiles_step<NTHR, QQ, D>>())             \
            return rc_;                                                        \
        hipLaunchKernelGGL((k_tiles_step<NTHR, QQ, D>),                       \
                           dim3(order_grid),                                   \
                           dim3(NTHR), lds, s, M, P->tile_rows, P->panels,     \
                           P->shift, p, order_arg, xr, P->cb,                  \
                           P->cpanel, P->nbk, P->ent, P->val, x, y);           \
    } while (0)
#define ST(NTHR, QQ)                                                           \
    do {                                                                       \
        if (P->det) {                                                          \
            if ((NTHR) <= 256)                                                 \
                ST_(256, 2, true);                                             \
            else                                                               \
                ST_(512, 2, true);                                             \
        } else                                                                 \
            ST_(NTHR, QQ, false);                                              \
    } while (0)
        if (variant & 2048) { /* tuning: two groups of 4 per lane */
            if (waves > 0 && waves < 8) ST(256, 2);
            else ST(512, 2);
        }
        else if (waves > 0 && waves < 8) ST(256, 1);
        else if (waves > 8) ST(1024, 1);
        else if (waves == 8) ST(512, 1);
        else if (per_bucket >= 3000.0) ST(512, 1);
        else ST(256, 1);
#undef ST
#undef ST_
    }
    return hip_errno(hipGetLastError());
}

/* sched: 0 = "steps", 1 = "sweep", 2 = "chain" (the steps layout, launched
 * as one chain launch), < 0 = the process default (spmv_set_panel_schedule /
 * SPMV_PANEL_SCHED); tile_rows: rows per tile of the steps layout (0 =
 * default, up to 16384) */
int panels_from_csr_opts(const spmv_csr_dev *A, const spmv_panel_opts *o,
                         spmv_panels **out) {
    return panels_build(A->M, A->N, A->NZ, o, 0, A->irp, NULL, 0, A->ja, A->as,
                        NULL, out);
}

int panels_from_hll_opts(const spmv_hll_dev *H, const spmv_panel_opts *o,
                         spmv_panels **out) {
    if (H->slots > 0 && !H->padmask)
        return -ENODATA;
    return panels_build(H->M, H->N, H->slots, o, H->nb, NULL, H->off,
                        H->col_major, H->ja, H->as, H->padmask, out);
}

extern "C" void spmv_panel_opts_default(spmv_panel_opts *o) {
    if (!o)
        return;
    memset(o, 0, sizeof *o);
    o->struct_size = (int)sizeof *o;
    o->sched = -1;
    o->sweep_layout = -1;
}

int panels_from_csr(const spmv_csr_dev *A, int panel_cols, int sched,
                    int tile_rows, spmv_panels **out) {
    spmv_panel_opts o;
    spmv_panel_opts_default(&o);
    o.sched = sched;
    o.panel_cols = panel_cols;
    o.tile_rows = tile_rows;
    return panels_from_csr_opts(A, &o, out);
}

int panels_from_hll(const spmv_hll_dev *H, int panel_cols, int sched,
                    int tile_rows, spmv_panels **out) {
    spmv_panel_opts o;
    spmv_panel_opts_default(&o);
    o.sched = sched;
    o.panel_cols = panel_cols;
    o.tile_rows = tile_rows;
    return panels_from_hll_opts(H, &o, out);
}

/* the options a copy was built with (build_panels_like) */
void panels_get_opts(const spmv_panels *P, spmv_panel_opts *o) {
    spmv_panel_opts_default(o);
    o->sched = P->sweep ? 1 : P->chain ? 2 : 0;
    /* the width actually used (a copy built with an explicit panel_cols is
     * rebuilt with it, not with the default 2^18) */
    o->panel_cols = 1 << P->shift;
    o->tile_rows = P->sweep ? 0 : P->tile_rows;
    o->sweep_wgs_per_cu = P->sweep ? P->wgs_per_cu : 0;
    o->reserve_cus = P->reserve_cus;
    o->lds_min = P->lds_min;
    o->tile_order = P->order;
    o->sweep_layout = P->pmajor;
    o->bucket_order = P->bucket_order;
    o->deterministic = P->det ? 1 : 2; /* explicit: a rebuild repeats it */
}

/* one-line description of a blocked copy: schedule, geometry, launch shape
 * (bench.py prints it; profiles are keyed on it) */
int panels_describe(const spmv_panels *P, char *buf, size_t len) {
    if (!P || !buf || !len)
        return -EINVAL;
    if (P->sweep)
        snprintf(buf, len,
                 "sweep tiles=%d x %d rows, panels=%d x 2^%d cols, grid=%d "
                 "(%d wg/cu, reserve %d), buckets %s-major, waves=%d",
                 P->tiles, P->tile_rows, P->panels, P->shift, P->grid,
                 P->wgs_per_cu, P->reserve_cus, P->pmajor ? "panel" : "tile",
                 P->waves_hint);
    else
        snprintf(buf, len,
                 "%s tiles=%d x %d rows, panels=%d x 2^%d cols, <=%d buckets "
                 "per tile (span %d, %s order), tile order %d, waves=%d",
                 P->chain ? "chain" : "steps", P->tiles, P->tile_rows,
                 P->panels, P->shift, P->max_nbk, P->span,
                 P->residue ? "residue" : "ascending", P->order,
                 P->waves_hint);
    if (P->det) {
        const size_t at = strlen(buf);
        snprintf(buf + at, len - at, ", deterministic");
    }
    if (P->nlong) {
        const size_t at = strlen(buf);
        snprintf(buf + at, len - at,
                 "; %d long row(s) beside the copy (%lld entries, %d segments)",
                 P->nlong, (long long)P->long_nnz, P->nlong_seg);
    }
    return 0;
}

int panels_is_sweep(const spmv_panels *P) { return P ? P->sweep : 0; }
int panels_is_chain(const spmv_panels *P) { return P ? P->chain : 0; }
void panels_set_chain(spmv_panels *P, int chain) {
    if (P && !P->sweep)
        P->chain = chain != 0;
}
int panels_waves(const spmv_panels *P) { return P ? P->waves_hint : 0; }
void panels_set_order(spmv_panels *P, int order) {
    if (P && !P->sweep && order >= 0 && order <= 2)
        P->order = order;
}
void panels_set_waves(spmv_panels *P, int waves) {
    if (P)
        P->waves_hint = waves > 0 ? waves : 0;
}
int panels_tile_rows(const spmv_panels *P) { return P ? P->tile_rows : 0; }

/* test hook (spmv_*_debug_stale_arrivals): what a launch that never
 * completed leaves in the long rows' arrival counters */
int panels_debug_stale_arrivals(spmv_panels *P) {
    if (!P || !P->nlong)
        return 0;
    HIP_RET(hipMemset(P->long_cnt, 0x01,
                      (size_t)P->nlong * sizeof(unsigned long long)));
    return P->nlong;
}

/* entries the copy stands for: in the buckets + in the long rows beside it */
int64_t panels_nnz(const spmv_panels *P) {
    return P ? P->nnz + P->long_nnz : 0;
}
int panels_count(const spmv_panels *P) { return P ? P->panels : 0; }
int panels_steps(const spmv_panels *P) {
    return P ? (P->sweep || P->chain ? 1 : P->max_nbk) : 0;
}
int panels_tiles(const spmv_panels *P) { return P ? P->tiles : 0; }

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
 * tile is 1024 consecutive rows (its slice of y fits LDS, 8 KiB), a
 * panel is 2^18 columns (2 MiB of x).  Entries are stored tile-major and,
 * inside a tile, in panel order (col int32, row-in-tile uint16, value f64:
 * 14 B).  Products are added into the LDS tile with the hardware LDS fp64
 * atomic (ds_add_f64): no segmented reduction, no ordering inside a bucket.
 *
 * Default schedule: ONE LAUNCH PER STEP; in step s workgroup t adds the
 * s-th non-empty bucket of tile t into its y slice (coalesced read-modify-write through LDS; a
 * tile has one owner per launch, launches are stream-ordered, y is zeroed
 * first).  Every CU gathers from the same panel by construction.  Cost:
 * y is re-read and re-written once per panel that touches the tile.
 * Measured on config 3 with columns anywhere: 2.62 ms vs 5.81 ms for the
 * direct kernels (2.2x).  It LOSES on matrices with locality (most
 * (tile, panel) buckets empty -> few workgroups per launch), so it is
 * opt-in and bench.py picks it only when it measures faster.
 *
 * Variant bit 3: one persistent launch, a workgroup keeps its tile in LDS
 * across all panels and writes y once (least traffic).  Workgroups drift
 * apart (no phase barrier), the L2 hit rate of the gathers falls to ~30 %
 * (rocprofv3 TCC_HIT/TCC_MISS) and it measures 4.3 ms; with all gathers
 * folded into one panel the same kernel runs 2.2 ms, which is what a
 * cheap-enough phase barrier could buy (next round).
 *
 * Summation order inside a row depends on LDS atomic arrival order: results
 * are reproducible to rounding (tests hold them to 1e-12 of the row scale),
 * not bitwise.
 */
#include <hipcub/hipcub.hpp>

#include "hip_common.h"

#define TILE_ROWS_MAX 8192 /* 64 KiB of LDS per workgroup */
#define TILE_THREADS 512

struct spmv_panels {
    int shift;       /* log2(columns per panel) */
    int panels;      /* column panels */
    int tile_rows;   /* rows per tile (multiple of 32) */
    int tiles;       /* row tiles */
    int64_t nnz;     /* entries kept */
    int *col;        /* [nnz] absolute column */
    unsigned short *rloc; /* [nnz] row inside its tile */
    double *val;     /* [nnz] */
    int64_t *tptr;   /* DEVICE [tiles+1] entry range of each tile */
    int64_t *bptr;   /* DEVICE [tiles*panels+1] start of bucket (tile, panel) */
    int64_t *cb;     /* DEVICE [tiles*panels*2] (begin,end) of the s-th NON-EMPTY
                        bucket of each tile, tile-major */
    int *nbk;        /* DEVICE [tiles] non-empty buckets per tile */
    int max_nbk;     /* launches needed = max over tiles */
};

void panels_free(spmv_panels *p) {
    if (!p)
        return;
    (void)hipFree(p->col);
    (void)hipFree(p->rloc);
    (void)hipFree(p->val);
    (void)hipFree(p->tptr);
    (void)hipFree(p->bptr);
    (void)hipFree(p->cb);
    (void)hipFree(p->nbk);
    free(p);
}

/* ---- keys: (tile * panels + panel), all-ones = dropped slot ---- */
__global__ void k_keys_from_csr(int M, int tile_rows, int panels, int shift,
                                const int *__restrict__ irp,
                                const int *__restrict__ ja, unsigned *key,
                                unsigned *idx) {
    /* 8 lanes per row keep the writes reasonably coalesced */
    const int sub = threadIdx.x & 7;
    long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    if (row >= M)
        return;
    const unsigned tk = (unsigned)(row / tile_rows) * (unsigned)panels;
    for (int k = irp[row] + sub, e = irp[row + 1]; k < e; k += 8) {
        key[k] = tk + (unsigned)(ja[k] >> shift);
        idx[k] = (unsigned)k;
    }
}

__global__ void k_keys_from_hll(int M, int tile_rows, int panels, int shift,
                                int col_major, const int64_t *__restrict__ off,
                                const int *__restrict__ ja,
                                const double *__restrict__ as, unsigned *key,
                                unsigned *idx) {
    int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= M)
        return;
    int b = row >> 5, i = row & 31;
    int rows = min(32, M - b * 32);
    int64_t o = off[b];
    int w = (int)((unsigned)(off[b + 1] - o) / (unsigned)rows);
    const unsigned tk = (unsigned)(row / tile_rows) * (unsigned)panels;
    for (int j = 0; j < w; ++j) {
        int64_t t = o + (col_major ? (int64_t)j * rows + i : (int64_t)i * w + j);
        /* pads carry the value 0.0 (hip_hll.h); a slot that is exactly zero
         * contributes nothing and is dropped */
        key[t] = as[t] != 0.0 ? tk + (unsigned)(ja[t] >> shift) : ~0u;
        idx[t] = (unsigned)t;
    }
}

/* row of a source position: CSR needs a search in irp, HLL decodes the slot */
__global__ void k_tile_gather_csr(int64_t n, int M, int tile_rows,
                                  const unsigned *__restrict__ idx,
                                  const int *__restrict__ irp,
                                  const int *__restrict__ ja,
                                  const double *__restrict__ as, int *tcol,
                                  unsigned short *trow, double *tval) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    const unsigned t = idx[k];
    int lo = 0, hi = M; /* last row with irp[row] <= t */
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if ((unsigned)irp[mid] <= t)
            lo = mid;
        else
            hi = mid;
    }
    tcol[k] = ja[t];
    trow[k] = (unsigned short)(lo % tile_rows);
    tval[k] = as[t];
}

__global__ void k_tile_gather_hll(int64_t n, int M, int nb, int tile_rows,
                                  int col_major,
                                  const unsigned *__restrict__ idx,
                                  const int64_t *__restrict__ off,
                                  const int *__restrict__ ja,
                                  const double *__restrict__ as, int *tcol,
                                  unsigned short *trow, double *tval) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    const unsigned t = idx[k];
    int lo = 0, hi = nb; /* block holding slot t */
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (off[mid] <= (int64_t)t)
            lo = mid;
        else
            hi = mid;
    }
    const int rows = min(32, M - lo * 32);
    const unsigned rel = (unsigned)((int64_t)t - off[lo]);
    const unsigned w = (unsigned)(off[lo + 1] - off[lo]) / (unsigned)rows;
    const int i = col_major ? (int)(rel % (unsigned)rows) : (int)(rel / w);
    tcol[k] = ja[t];
    trow[k] = (unsigned short)((lo * 32 + i) % tile_rows);
    tval[k] = as[t];
}

/* first position whose key >= tile * panels, one thread per tile boundary */
__global__ void k_tile_bounds(int tiles, int panels, int64_t n,
                              const unsigned *__restrict__ key, int64_t *tptr) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > tiles)
        return;
    const uint64_t bound = (uint64_t)t * (uint64_t)panels;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((uint64_t)key[mid] < bound)
            lo = mid + 1;
        else
            hi = mid;
    }
    tptr[t] = lo;
}

/* start of every (tile, panel) bucket: first position with key >= b */
__global__ void k_bucket_bounds(int64_t buckets, int64_t n,
                                const unsigned *__restrict__ key,
                                int64_t *bptr) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > buckets)
        return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((uint64_t)key[mid] < (uint64_t)b)
            lo = mid + 1;
        else
            hi = mid;
    }
    bptr[b] = lo;
}

/* compact every tile's non-empty buckets; one thread per tile */
__global__ void k_compact_buckets(int tiles, int panels,
                                  const int64_t *__restrict__ bptr,
                                  int64_t *cb, int *nbk) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles)
        return;
    int n = 0;
    for (int p = 0; p < panels; ++p) {
        int64_t b = bptr[(int64_t)t * panels + p];
        int64_t e = bptr[(int64_t)t * panels + p + 1];
        if (e > b) {
            cb[((int64_t)t * panels + n) * 2] = b;
            cb[((int64_t)t * panels + n) * 2 + 1] = e;
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

static int panels_build(int M, int N, int64_t slots, int panel_cols, int nb,
                        const int *irp_or_null, const int64_t *off_or_null,
                        int col_major, const int *ja, const double *as,
                        spmv_panels **out) {
    int rc = 0;
    *out = NULL;
    if (slots > (int64_t)INT32_MAX)
        return -EOVERFLOW;
    int shift = 18; /* 2^18 columns = 2 MiB of x: half of an XCD's L2 */
    if (panel_cols > 0) {
        shift = 0;
        while ((1 << (shift + 1)) <= panel_cols && shift < 30)
            ++shift;
    }
    const int panels = (int)((((int64_t)(N > 0 ? N : 1) - 1) >> shift) + 1);
    /* tile height: 1024 rows (8 KiB of LDS) measured best for the default
     * one-launch-per-panel schedule on config 3 (2.62 ms; 3.3 ms at 8192):
     * many small workgroups overlap their load / gather / store phases */
    long long tr = 1024;
    if (const char *ev = getenv("SPMV_TILE_ROWS")) { /* tuning override */
        long long o = atoll(ev);
        if (o >= 32 && o <= TILE_ROWS_MAX)
            tr = o / 32 * 32;
    }
    const int tiles = (int)(((long long)M + tr - 1) / tr);
    if ((uint64_t)(tiles > 0 ? tiles : 1) * (uint64_t)panels >= 0xffffffffull)
        return -EOVERFLOW;

    spmv_panels *P = (spmv_panels *)calloc(1, sizeof *P);
    if (!P)
        return -ENOMEM;
    P->shift = shift;
    P->panels = panels;
    P->tile_rows = (int)tr;
    P->tiles = tiles;
    unsigned *key[2] = {NULL, NULL};
    unsigned *idx[2] = {NULL, NULL};
    unsigned *skey = NULL, *sidx = NULL;
    void *tmp = NULL;
    size_t tmp_bytes = 0;
    const size_t n = (size_t)(slots > 0 ? slots : 1);
    int64_t total = 0;

    for (int k = 0; k < 2; ++k) {
        HIP_TRY(hipMalloc((void **)&key[k], n * sizeof(unsigned)));
        HIP_TRY(hipMalloc((void **)&idx[k], n * sizeof(unsigned)));
    }
    skey = key[0];
    sidx = idx[0];
    if (slots > 0) {
        if (irp_or_null)
            hipLaunchKernelGGL(k_keys_from_csr,
                               dim3((unsigned)(((long long)M * 8 + 255) / 256)),
                               dim3(256), 0, 0, M, (int)tr, panels, shift,
                               irp_or_null, ja, key[0], idx[0]);
        else
            hipLaunchKernelGGL(k_keys_from_hll, dim3((M + 255) / 256),
                               dim3(256), 0, 0, M, (int)tr, panels, shift,
                               col_major, off_or_null, ja, as, key[0], idx[0]);
        HIP_TRY(hipGetLastError());
        {
            hipcub::DoubleBuffer<unsigned> dk(key[0], key[1]);
            hipcub::DoubleBuffer<unsigned> dv(idx[0], idx[1]);
            /* CSR sources have no dropped slots: sort only the used bits */
            int end_bit = 32;
            if (irp_or_null) {
                end_bit = 1;
                while (end_bit < 32 &&
                       (((uint64_t)tiles * (uint64_t)panels) >> end_bit))
                    ++end_bit;
            }
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(
                NULL, tmp_bytes, dk, dv, (int)slots, 0, end_bit, 0));
            HIP_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(
                tmp, tmp_bytes, dk, dv, (int)slots, 0, end_bit, 0));
            HIP_TRY(hipDeviceSynchronize());
            skey = dk.Current();
            sidx = dv.Current();
        }
    }
    HIP_TRY(hipMalloc((void **)&P->tptr, ((size_t)tiles + 1) * sizeof(int64_t)));
    hipLaunchKernelGGL(k_tile_bounds, dim3((tiles + 256) / 256), dim3(256), 0,
                       0, tiles, panels, slots, skey, P->tptr);
    HIP_TRY(hipGetLastError());
    {
        const int64_t buckets = (int64_t)tiles * panels;
        HIP_TRY(hipMalloc((void **)&P->bptr,
                          ((size_t)buckets + 1) * sizeof(int64_t)));
        hipLaunchKernelGGL(k_bucket_bounds,
                           dim3((unsigned)((buckets + 256) / 256)), dim3(256),
                           0, 0, buckets, slots, skey, P->bptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMalloc((void **)&P->cb,
                          ((size_t)buckets + 1) * 2 * sizeof(int64_t)));
        HIP_TRY(hipMalloc((void **)&P->nbk, ((size_t)tiles + 1) * sizeof(int)));
        HIP_TRY(hipMemset(P->nbk + tiles, 0, sizeof(int)));
        if (tiles > 0) {
            hipLaunchKernelGGL(k_compact_buckets, dim3((tiles + 255) / 256),
                               dim3(256), 0, 0, tiles, panels, P->bptr, P->cb,
                               P->nbk);
            hipLaunchKernelGGL(k_max_int, dim3(64), dim3(256), 0, 0, tiles,
                               P->nbk, P->nbk + tiles);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipMemcpy(&P->max_nbk, P->nbk + tiles, sizeof(int),
                          hipMemcpyDeviceToHost));
    }
    HIP_TRY(hipMemcpy(&total, P->tptr + tiles, sizeof total,
                      hipMemcpyDeviceToHost));
    P->nnz = total; /* dropped slots sort behind the last tile */
    {
        const size_t m = (size_t)(P->nnz > 0 ? P->nnz : 1);
        HIP_TRY(hipMalloc((void **)&P->col, m * sizeof(int)));
        HIP_TRY(hipMalloc((void **)&P->rloc, m * sizeof(unsigned short)));
        HIP_TRY(hipMalloc((void **)&P->val, m * sizeof(double)));
    }
    if (P->nnz > 0) {
        const unsigned g = (unsigned)((P->nnz + 255) / 256);
        if (irp_or_null)
            hipLaunchKernelGGL(k_tile_gather_csr, dim3(g), dim3(256), 0, 0,
                               P->nnz, M, (int)tr, sidx, irp_or_null, ja, as,
                               P->col, P->rloc, P->val);
        else
            hipLaunchKernelGGL(k_tile_gather_hll, dim3(g), dim3(256), 0, 0,
                               P->nnz, M, nb, (int)tr, col_major, sidx,
                               off_or_null, ja, as, P->col, P->rloc, P->val);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
    }
    *out = P;
    P = NULL;
fail:
    for (int k = 0; k < 2; ++k) {
        (void)hipFree(key[k]);
        (void)hipFree(idx[k]);
    }
    (void)hipFree(tmp);
    panels_free(P);
    return rc;
}

/* ------------------------------------------------------------------ */
/* the kernel: persistent workgroups, one row tile at a time             */
/* ------------------------------------------------------------------ */
#define TILE_UNROLL 8

template <int ABL> /* ablation bits for timing experiments: 1 = no LDS add,
                      2 = all gathers folded into one panel, 4 = no x gather */
__global__ void __launch_bounds__(TILE_THREADS)
    k_tiles_spmv(int M, int tile_rows, int tiles,
                 const int64_t *__restrict__ tptr, const int *__restrict__ tcol,
                 const unsigned short *__restrict__ trow,
                 const double *__restrict__ tval, const double *__restrict__ x,
                 double *__restrict__ y) {
    extern __shared__ double ytile[];
    const int tid = threadIdx.x;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        for (int i = tid; i < tile_rows; i += TILE_THREADS)
            ytile[i] = 0.0;
        __syncthreads();
        const int64_t b = tptr[t], e = tptr[t + 1];
        for (int64_t k0 = b + tid; k0 < e; k0 += TILE_THREADS * TILE_UNROLL) {
            int c[TILE_UNROLL];
            unsigned short rl[TILE_UNROLL];
            double v[TILE_UNROLL], xv[TILE_UNROLL];
#pragma unroll
            for (int u = 0; u < TILE_UNROLL; ++u) {
                const int64_t k = k0 + (int64_t)u * TILE_THREADS;
                const bool ok = k < e;
                c[u] = ok ? __builtin_nontemporal_load(tcol + k) : -1;
                rl[u] = ok ? __builtin_nontemporal_load(trow + k) : 0;
                v[u] = ok ? __builtin_nontemporal_load(tval + k) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < TILE_UNROLL; ++u)
                xv[u] = c[u] >= 0 ? ((ABL & 4) ? 1.0
                                     : (ABL & 2) ? x[c[u] & 0x3FFFF] /* one panel */
                                                 : x[c[u]])
                                  : 0.0;
#pragma unroll
            for (int u = 0; u < TILE_UNROLL; ++u) {
                if (ABL & 1) {
                    if (v[u] * xv[u] == 1.2345e300)
                        ytile[rl[u]] = 1.0;
                } else if (c[u] >= 0) {
                    unsafeAtomicAdd(&ytile[rl[u]], v[u] * xv[u]);
                }
            }
        }
        __syncthreads();
        const int64_t row0 = (int64_t)t * tile_rows;
        for (int i = tid; i < tile_rows && row0 + i < M; i += TILE_THREADS)
            y[row0 + i] = ytile[i];
        __syncthreads();
    }
}

/*
 * Variant with a hard phase boundary: one launch per column panel.  Every
 * workgroup adds its (tile, panel) bucket into the y slice of its tile
 * (through LDS, coalesced read-modify-write; the tile has a single owner per
 * launch and launches are stream-ordered).  All CUs gather from one panel by
 * construction, at the price of re-reading and re-writing y once per panel.
 */
template <int NT, int ABL = 0, int UN = TILE_UNROLL> /* timing ablations: 1 no y traffic, 2 no gather */
__global__ void __launch_bounds__(NT)
    k_tiles_one_panel(int M, int tile_rows, int panels, int step,
                      const int64_t *__restrict__ cb,
                      const int *__restrict__ nbk,
                      const int *__restrict__ tcol,
                      const unsigned short *__restrict__ trow,
                      const double *__restrict__ tval,
                      const double *__restrict__ x, double *__restrict__ y) {
    extern __shared__ double ytile[];
    const int tid = threadIdx.x;
    /* XCD-contiguous tile ranges: the tiles an XCD runs at one time are
     * neighbours, so their step-th panels coincide or are adjacent */
    int t;
    {
        const int nx = 8, nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk / nx, r = nblk % nx, xx = bid % nx, kk = bid / nx;
        t = xx * q + (xx < r ? xx : r) + kk;
    }
    if (step >= nbk[t])
        return; /* this tile has fewer non-empty buckets: y untouched */
    const int64_t b = cb[((int64_t)t * panels + step) * 2];
    const int64_t e = cb[((int64_t)t * panels + step) * 2 + 1];
    const int64_t row0 = (int64_t)t * tile_rows;
    /* first batch of entries and the y slice are fetched together (they are
     * independent); the x gathers follow the barrier: issued before it they
     * delay the slice, whose loads return in order behind them (measured
     * 2.97 ms vs 2.6 ms) */
    int c[UN];
    unsigned short rl[UN];
    double v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const int64_t k = b + tid + (int64_t)u * NT;
        const bool ok = k < e;
        c[u] = ok ? __builtin_nontemporal_load(tcol + k) : -1;
        rl[u] = ok ? __builtin_nontemporal_load(trow + k) : 0;
        v[u] = ok ? __builtin_nontemporal_load(tval + k) : 0.0;
    }
    for (int i = tid; i < tile_rows; i += NT)
        ytile[i] = (!(ABL & 1) && row0 + i < M) ? y[row0 + i] : 0.0;
    __syncthreads();
    for (int64_t k0 = b + tid; k0 < e; k0 += (int64_t)NT * UN) {
        double xv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u)
            xv[u] = c[u] >= 0 ? ((ABL & 2) ? x[c[u] & 1023] : x[c[u]]) : 0.0;
        double pr[UN];
        unsigned short rr[UN];
        bool on[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            pr[u] = v[u] * xv[u];
            rr[u] = rl[u];
            on[u] = c[u] >= 0;
        }
        if (k0 + (int64_t)NT * UN < e) { /* next batch behind the gathers */
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int64_t k = k0 + (int64_t)NT * UN + (int64_t)u * NT;
                const bool ok = k < e;
                c[u] = ok ? __builtin_nontemporal_load(tcol + k) : -1;
                rl[u] = ok ? __builtin_nontemporal_load(trow + k) : 0;
                v[u] = ok ? __builtin_nontemporal_load(tval + k) : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (on[u])
                unsafeAtomicAdd(&ytile[rr[u]], pr[u]);
    }
    __syncthreads();
    if (ABL & 1) {
        if (ytile[tid] == 1.2345e300)
            y[row0] = 1.0;
        return;
    }
    for (int i = tid; i < tile_rows && row0 + i < M; i += NT)
        y[row0 + i] = ytile[i];
}

int panels_launch(const spmv_panels *P, int M, int waves, int variant,
                  const double *x, double *y, hipStream_t s) {
    if (!P)
        return -EINVAL;
    if (M == 0)
        return 0;
    hipDeviceProp_t prop;
    int dev = 0;
    HIP_RET(hipGetDevice(&dev));
    HIP_RET(hipGetDeviceProperties(&prop, dev));
    int grid = prop.multiProcessorCount * 2;
    if (grid > P->tiles)
        grid = P->tiles;
    const size_t lds = (size_t)P->tile_rows * sizeof(double);
    if (!(variant & 8)) { /* default: one launch per panel */
        HIP_RET(hipMemsetAsync(y, 0, (size_t)M * sizeof(double), s));
        const bool small = (waves > 0 && waves < 8);
        /* launch `step` handles the step-th NON-EMPTY bucket of every tile:
         * a matrix whose rows reach over k panels needs k launches, all
         * tiles busy in each of them */
        const int abl = (variant >> 4) & 3;
        const int un = (variant >> 8) & 15; /* tuning: unroll override */
        for (int p = 0; p < P->max_nbk; ++p) {
#define TP(NTHR, UNR)                                                          \
    hipLaunchKernelGGL((k_tiles_one_panel<NTHR, 0, UNR>), dim3(P->tiles),     \
                       dim3(NTHR), lds, s, M, P->tile_rows, P->panels, p,     \
                       P->cb, P->nbk, P->col, P->rloc, P->val, x, y)
            if (un == 1) { if (small) TP(256, 1); else TP(512, 1); continue; }
            if (un == 2) { if (small) TP(256, 2); else TP(512, 2); continue; }
            if (un == 4) { if (small) TP(256, 4); else TP(512, 4); continue; }
#undef TP
            if (abl == 1)
                hipLaunchKernelGGL((k_tiles_one_panel<512, 1>), dim3(P->tiles),
                                   dim3(512), lds, s, M, P->tile_rows,
                                   P->panels, p, P->cb, P->nbk, P->col,
                                   P->rloc, P->val, x, y);
            else if (abl == 2)
                hipLaunchKernelGGL((k_tiles_one_panel<512, 2>), dim3(P->tiles),
                                   dim3(512), lds, s, M, P->tile_rows,
                                   P->panels, p, P->cb, P->nbk, P->col,
                                   P->rloc, P->val, x, y);
            else if (abl == 3)
                hipLaunchKernelGGL((k_tiles_one_panel<512, 3>), dim3(P->tiles),
                                   dim3(512), lds, s, M, P->tile_rows,
                                   P->panels, p, P->cb, P->nbk, P->col,
                                   P->rloc, P->val, x, y);
            else if (small)
                hipLaunchKernelGGL((k_tiles_one_panel<256, 0, 4>),
                                   dim3(P->tiles), dim3(256), lds, s, M,
                                   P->tile_rows, P->panels, p, P->cb, P->nbk,
                                   P->col, P->rloc, P->val, x, y);
            else /* 512 lanes x 4 entries: 2.58 ms (x8: 2.65) on config 3 */
                hipLaunchKernelGGL((k_tiles_one_panel<512, 0, 4>),
                                   dim3(P->tiles), dim3(512), lds, s, M,
                                   P->tile_rows, P->panels, p, P->cb, P->nbk,
                                   P->col, P->rloc, P->val, x, y);
        }
        return hip_errno(hipGetLastError());
    }
#define TL(A) hipLaunchKernelGGL(k_tiles_spmv<A>, dim3(grid), dim3(TILE_THREADS), \
                                 lds, s, M, P->tile_rows, P->tiles, P->tptr,     \
                                 P->col, P->rloc, P->val, x, y)
    switch ((variant >> 4) & 7) {
    case 1: TL(1); break;
    case 2: TL(2); break;
    case 4: TL(4); break;
    case 5: TL(5); break;
    default: TL(0); break;
    }
#undef TL
    return hip_errno(hipGetLastError());
}

int panels_from_csr(const spmv_csr_dev *A, int panel_cols, spmv_panels **out) {
    return panels_build(A->M, A->N, A->NZ, panel_cols, 0, A->irp, NULL, 0,
                        A->ja, A->as, out);
}

int panels_from_hll(const spmv_hll_dev *H, int panel_cols, spmv_panels **out) {
    return panels_build(H->M, H->N, H->slots, panel_cols, H->nb, NULL, H->off,
                        H->col_major, H->ja, H->as, out);
}

int64_t panels_nnz(const spmv_panels *P) { return P ? P->nnz : 0; }
int panels_count(const spmv_panels *P) { return P ? P->panels : 0; }
int panels_steps(const spmv_panels *P) { return P ? P->max_nbk : 0; }
int panels_tiles(const spmv_panels *P) { return P ? P->tiles : 0; }

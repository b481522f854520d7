/*
 * hll_kernels.hip -- fp64 HLL (hacked ELLPACK, hack = 32) SpMV kernels for
 * gfx950 (wave64).
 *
 * Four kernels fill the four slots of the reference's driver table
 * (reference cuda_hll.cu:19-152, main.c:310-315); the designs are new.
 * Device layout: one JA slab and one AS slab; hack block b occupies slots
 * [off[b], off[b+1]) with rows_b = min(32, M - 32 b) rows and width
 * w_b = (off[b+1] - off[b]) / rows_b; pads already point at a valid column
 * and carry the value 0.0 (hip_hll.h).
 *
 *   0 threads_row_major  lane per row; slot(i,j) = i*w + j.
 *   1 threads_col_major  lane per row; a 64-lane wavefront owns TWO hack
 *                        blocks (lanes 0-31 / 32-63).  Each block's slab is
 *                        contiguous, so the wavefront copies it to LDS in
 *                        chunks of 8 columns with 16 B/lane loads (1 KiB per
 *                        instruction, fully coalesced), then every lane
 *                        walks its row out of LDS (conflict-free: lane i
 *                        reads word j*32+i) and gathers x.
 *   2 wave_block         same mapping, direct global loads, no LDS.
 *   3 subwave_row        16 lanes per row over row-major blocks,
 *                        __shfl_down(width 16) reduction.
 *
 * No MFMA (no dense contraction).  JA/AS stream once -> non-temporal loads;
 * x gathers are ordinary cached loads.
 */
#include <algorithm>
#include "hip_common.h"

#define HACK 32
/* grouped order: runs of this many consecutive workgroups per XCD, the runs
 * dealt to the XCDs round-robin (see panels.hip, the blocked schedules) */
#define HLL_GROUP 32

template <typename T> __device__ __forceinline__ T ld_stream(const T *p) {
    return __builtin_nontemporal_load(p);
}

typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

/* ------------------------------------------------------------------ */
/* pad rewrite on the device (reference cuda_hll.cu:173-195 does it on  */
/* the host, block by block): pad -> previous valid column, or 0.        */
/* ------------------------------------------------------------------ */
/* also records which slots were pads (bit t of padmask, zeroed by the    */
/* caller): the blocked copy drops exactly those, never an explicit zero */
__global__ void k_hll_fix_pads(int M, int col_major,
                               const int64_t *__restrict__ off,
                               int *__restrict__ ja,
                               unsigned *__restrict__ padmask) {
    int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= M)
        return;
    int b = row / HACK, i = row % HACK;
    int rows = min(HACK, M - b * HACK);
    int64_t o = off[b];
    int w = (int)((off[b + 1] - o) / rows);
    int last = 0;
    for (int j = 0; j < w; ++j) {
        int64_t t = o + (col_major ? (int64_t)j * rows + i : (int64_t)i * w + j);
        int c = ja[t];
        if (c < 0) {
            ja[t] = last;
            atomicOr(padmask + (t >> 5), 1u << (t & 31));
        } else {
            last = c;
        }
    }
}

int hll_fix_pads_dev(spmv_hll_dev *H, hipStream_t s) {
    if (H->M == 0)
        return 0;
    HIP_RET(hipMemsetAsync(H->padmask, 0,
                           (((size_t)H->slots + 31) / 32 + 1) * sizeof(unsigned),
                           s));
    hipLaunchKernelGGL(k_hll_fix_pads, dim3((H->M + 255) / 256), dim3(256), 0,
                       s, H->M, H->col_major, H->off, H->ja, H->padmask);
    return hip_errno(hipGetLastError());
}

/* ------------------------------------------------------------------ */
/* 0: lane per row, row-major                                           */
/* ------------------------------------------------------------------ */
/* `wide` > 0 (kernels 0-3): the matrix has hack blocks of more than `wide`
 * columns; they are left to k_hll_wide, launched right after on the same
 * stream */
__global__ void k_hll_row_major(int M, int b0, int b1, int wide,
                                const int64_t *__restrict__ off,
                                const int *__restrict__ ja,
                                const double *__restrict__ as,
                                const double *__restrict__ x,
                                double *__restrict__ y) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int b = b0 + (int)(t / HACK), i = (int)(t % HACK);
    if (b >= b1)
        return;
    int rows = min(HACK, M - b * HACK);
    if (i >= rows)
        return;
    int64_t o = off[b];
    int w = hack_block_width(off, b, rows);
    if (wide > 0 && w > wide)
        return;
    const int *rj = ja + o + (int64_t)i * w;
    const double *ra = as + o + (int64_t)i * w;
    /* the row's slots in order, four loads in flight (hip_common.h) */
    y[(int64_t)b * HACK + i] = strided_dot<1, 4>(rj, ra, x, 0, w, 0);
}

/* ------------------------------------------------------------------ */
/* 2: wavefront per pair of hack blocks, col-major, direct loads        */
/*    Software pipelined: while the x gathers of columns [j, j+U) are    */
/*    in flight the JA/AS loads of [j+U, j+2U) are already issued, so    */
/*    the stream never waits behind a gather round trip.                 */
/* ------------------------------------------------------------------ */
/*
 * Workgroup -> hack blocks.  REMAP: workgroups are dealt round-robin over the
 * 8 XCDs; XCD k runs the CONTIGUOUS block range [xr.first[k], xr.first[k+1])
 * so that neighbouring row tiles (which share their x window) meet in one L2
 * (+18 % on a band of 2048 columns), and the ranges hold about equal numbers
 * of SLOTS, not of blocks: a matrix whose rows are wider in one half
 * otherwise leaves XCDs idle (the finding of round 2 on the blocked path,
 * DESIGN.md section 4).  The launch has 8 x (longest range) workgroups; the
 * surplus ones of shorter ranges exit.
 */
template <int U, int ORDER, int ABL = 0> /* ABL 1: all gathers read x[0] */
__global__ void k_hll_col_direct(int M, int b0, int b1, int wide, xcd_ranges xr,
                                 const int64_t *__restrict__ off,
                                 const int *__restrict__ ja,
                                 const double *__restrict__ as,
                                 const double *__restrict__ x,
                                 double *__restrict__ y) {
    int b, i;
    if (ORDER == 1) {
        const int xx = blockIdx.x % NUM_XCD, kk = blockIdx.x / NUM_XCD;
        const long long t = (long long)kk * blockDim.x + threadIdx.x;
        b = xr.first[xx] + (int)(t / HACK);
        i = (int)(t % HACK);
        if (b >= xr.first[xx + 1])
            return;
    } else if (ORDER == 2) { /* groups of HLL_GROUP workgroups per XCD */
        const int xx = blockIdx.x % NUM_XCD, kk = blockIdx.x / NUM_XCD;
        const long long w =
            ((long long)(kk / HLL_GROUP) * NUM_XCD + xx) * HLL_GROUP + kk % HLL_GROUP;
        const long long t = w * blockDim.x + threadIdx.x;
        b = b0 + (int)(t / HACK);
        i = (int)(t % HACK);
    } else {
        const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        b = b0 + (int)(t / HACK);
        i = (int)(t % HACK);
    }
    if (b >= b1)
        return;
    int rows = min(HACK, M - b * HACK);
    if (i >= rows)
        return;
    const int64_t o = off[b];
    const int w = hack_block_width(off, b, rows);
    if (wide > 0 && w > wide)
        return;
    const int *cj = ja + o + i;
    const double *ca = as + o + i;
    double acc = 0.0;
    int cJ[U];
    double cA[U];
    const int nfull = w / U;
    if (nfull > 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cJ[u] = ld_stream(cj + u * rows);
            cA[u] = ld_stream(ca + u * rows);
        }
    }
    for (int c = 0; c < nfull; ++c) {
        double xv[U], av[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            xv[u] = (ABL & 1) ? x[cJ[u] & 1] : x[cJ[u]];
            av[u] = cA[u];
        }
        if (c + 1 < nfull) {
            const int *nj = cj + (size_t)(c + 1) * U * rows;
            const double *na = ca + (size_t)(c + 1) * U * rows;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                cJ[u] = ld_stream(nj + u * rows);
                cA[u] = ld_stream(na + u * rows);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            acc += av[u] * xv[u];
    }
    for (int j = nfull * U; j < w; ++j)
        acc += ld_stream(ca + (size_t)j * rows) *
               x[ld_stream(cj + (size_t)j * rows)];
    __builtin_nontemporal_store(acc, y + (int64_t)b * HACK + i);
}

/* ------------------------------------------------------------------ */
/* 1: lane per row, col-major, hack blocks staged through LDS           */
/*    (full 32-row blocks only; the launcher sends a ragged tail block  */
/*    to k_hll_col_direct)                                              */
/* ------------------------------------------------------------------ */
#define CH 8                     /* columns per staged chunk */
#define CH_SLOTS (CH * HACK)     /* 256 slots: 1 KiB of JA, 2 KiB of AS */

struct hll_chunk {
    v4i jA, jB;
    v2d aA0, aA1, aB0, aB1;
};

/* coalesced 16 B/lane loads of slots [s0, s0+256) of the wavefront's two
 * blocks (predicated past the end of each block) */
__device__ __forceinline__ void hll_chunk_load(hll_chunk &c, int s0, int lane,
                                               const int *gjA, const int *gjB,
                                               const double *gaA,
                                               const double *gaB, int nA,
                                               int nB) {
    const int sj = s0 + 4 * lane; /* 4 ints */
    const int sa = s0 + 2 * lane; /* 2 doubles, twice */
    const v4i zi = {0, 0, 0, 0};
    const v2d zd = {0, 0};
    c.jA = sj < nA ? ld_stream((const v4i *)(gjA + sj)) : zi;
    c.jB = sj < nB ? ld_stream((const v4i *)(gjB + sj)) : zi;
    c.aA0 = sa < nA ? ld_stream((const v2d *)(gaA + sa)) : zd;
    c.aA1 = sa + 128 < nA ? ld_stream((const v2d *)(gaA + sa + 128)) : zd;
    c.aB0 = sa < nB ? ld_stream((const v2d *)(gaB + sa)) : zd;
    c.aB1 = sa + 128 < nB ? ld_stream((const v2d *)(gaB + sa + 128)) : zd;
}

template <int ORDER>
__global__ void k_hll_col_lds(int b0, int b1, int wide, xcd_ranges xr,
                              const int64_t *__restrict__ off,
                              const int *__restrict__ ja,
                              const double *__restrict__ as,
                              const double *__restrict__ x,
                              double *__restrict__ y) {
    /* per wavefront: two blocks x (256 ints + 256 doubles) = 6 KiB */
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = threadIdx.x / WAVE;
    const int waves = blockDim.x / WAVE;
    double *s_as = (double *)lds_raw + (size_t)wave * 2 * CH_SLOTS;
    int *s_ja = (int *)((double *)lds_raw + (size_t)waves * 2 * CH_SLOTS) +
                (size_t)wave * 2 * CH_SLOTS;

    int bA; /* wave-uniform */
    if (ORDER == 1) {
        const int xx = blockIdx.x % NUM_XCD, kk = blockIdx.x / NUM_XCD;
        bA = xr.first[xx] + 2 * (kk * waves + wave);
        if (xr.first[xx + 1] < b1)
            b1 = xr.first[xx + 1]; /* the pair stays inside the XCD's range */
    } else if (ORDER == 2) { /* groups of HLL_GROUP workgroups per XCD */
        const int xx = blockIdx.x % NUM_XCD, kk = blockIdx.x / NUM_XCD;
        const int w = ((kk / HLL_GROUP) * NUM_XCD + xx) * HLL_GROUP + kk % HLL_GROUP;
        bA = b0 + 2 * (w * waves + wave);
    } else {
        bA = b0 + 2 * ((int)blockIdx.x * waves + wave);
    }
    if (bA >= b1)
        return;
    const bool hasB = bA + 1 < b1;
    const int64_t oA = off[bA], oB = off[bA + 1];
    /* slots of block A = 32 * wA; 64-bit until the wide blocks are out (a
     * hub block of 2^26 columns and more does not fit an int) */
    const int64_t nA64 = oB - oA, nB64 = hasB ? off[bA + 2] - oB : 0;
    /* a wide block of the pair is k_hll_wide's: nothing read, nothing stored */
    const bool skipA = wide > 0 && (nA64 >> 5) > wide;
    const bool skipB = wide > 0 && (nB64 >> 5) > wide;
    const int nA = skipA ? 0 : (int)nA64;
    const int nB = skipB ? 0 : (int)nB64;
    const int half = lane >> 5, i = lane & 31;
    const int w = (half ? nB : nA) >> 5;
    const int nmax = nA > nB ? nA : nB;

    const int *gjA = ja + oA, *gjB = ja + oB;
    const double *gaA = as + oA, *gaB = as + oB;
    const int *lj = s_ja + half * CH_SLOTS + i;
    const double *la = s_as + half * CH_SLOTS + i;
    double acc = 0.0;

    /* register-staged pipeline: chunk c+1 is in flight from HBM while chunk
     * c is consumed out of LDS */
    hll_chunk cur;
    hll_chunk_load(cur, 0, lane, gjA, gjB, gaA, gaB, nA, nB);
    for (int s0 = 0; s0 < nmax; s0 += CH_SLOTS) {
        *(v4i *)(s_ja + 4 * lane) = cur.jA;
        *(v4i *)(s_ja + CH_SLOTS + 4 * lane) = cur.jB;
        *(v2d *)(s_as + 2 * lane) = cur.aA0;
        *(v2d *)(s_as + 128 + 2 * lane) = cur.aA1;
        *(v2d *)(s_as + CH_SLOTS + 2 * lane) = cur.aB0;
        *(v2d *)(s_as + CH_SLOTS + 128 + 2 * lane) = cur.aB1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        /* this lane's 8 columns out of LDS, their x gathers issued first */
        const int c0 = s0 >> 5;
        int cc[CH];
        double av[CH], xv[CH];
#pragma unroll
        for (int jj = 0; jj < CH; ++jj) {
            cc[jj] = lj[jj * HACK];
            av[jj] = la[jj * HACK];
        }
#pragma unroll
        for (int jj = 0; jj < CH; ++jj)
            xv[jj] = (c0 + jj < w) ? x[cc[jj]] : 0.0;
        if (s0 + CH_SLOTS < nmax)
            hll_chunk_load(cur, s0 + CH_SLOTS, lane, gjA, gjB, gaA, gaB, nA,
                           nB);
#pragma unroll
        for (int jj = 0; jj < CH; ++jj)
            if (c0 + jj < w)
                acc += av[jj] * xv[jj];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (half == 0 ? !skipA : (hasB && !skipB))
        __builtin_nontemporal_store(acc, y + (int64_t)(bA + half) * HACK + i);
}

/* ------------------------------------------------------------------ */
/* 3: 16 lanes per row, row-major                                       */
/* ------------------------------------------------------------------ */
__global__ void k_hll_subwave_row(int M, int b0, int b1, int wide,
                                  const int64_t *__restrict__ off,
                                  const int *__restrict__ ja,
                                  const double *__restrict__ as,
                                  const double *__restrict__ x,
                                  double *__restrict__ y) {
    const int sub = threadIdx.x & 15;
    /* grid-stride over the rows: 16 work-items per row are more than one
     * launch holds (2^32) beyond 2^28 rows.  The trip count is the same for
     * every lane of a wavefront (`first` is the wavefront's first row); rows
     * past the end are masked */
    const long long total = (long long)(b1 - b0) * HACK;
    const long long step = ((long long)gridDim.x * blockDim.x) >> 4;
    const long long lane_g = (threadIdx.x & (WAVE - 1)) >> 4;
    for (long long first =
             (((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4) - lane_g;
         first < total; first += step) {
        const long long g = first + lane_g;
        const int b = b0 + (int)(g / HACK), i = (int)(g % HACK);
        double acc = 0.0;
        bool live = false;
        if (g < total) {
            int rows = min(HACK, M - b * HACK);
            if (i < rows) {
                int64_t o = off[b];
                int w = hack_block_width(off, b, rows);
                live = !(wide > 0 && w > wide);
                const int *rj = ja + o + (int64_t)i * w;
                const double *ra = as + o + (int64_t)i * w;
                if (live)
                    acc = strided_dot<16, 4>(rj, ra, x, 0, w, sub);
            }
        }
#pragma unroll
        for (int d = 8; d > 0; d >>= 1)
            acc += __shfl_down(acc, d, 16);
        if (live && sub == 0)
            y[(int64_t)b * HACK + i] = acc;
    }
}

/* ------------------------------------------------------------------ */
/* wide hack blocks (> HLL_WIDE columns): workgroup g sums one segment of  */
/* columns (seg.y wide) of block b for all its rows, either layout; the    */
/* block's last segment to arrive adds the partial row sums in a fixed      */
/* order (deterministic), writes y and re-arms the counter.                */
/* ------------------------------------------------------------------ */
__global__ void __launch_bounds__(256)
    k_hll_wide(int M, int b0, int b1, int col_major,
               const int4 *__restrict__ seg, const int64_t *__restrict__ off,
               const int *__restrict__ ja, const double *__restrict__ as,
               const double *__restrict__ x, double *__restrict__ y,
               double *part, unsigned long long *cnt, unsigned epoch) {
    __shared__ double red[8][HACK];
    __shared__ int s_seen;
    const int g = blockIdx.x, tid = threadIdx.x;
    const int4 sg = seg[g];
    const int b = sg.x, segw = sg.y, kseg = sg.z, nseg = sg.w;
    if (b < b0 || b >= b1)
        return; /* another launch of a chunked exchange owns this block */
    const int rows = min(HACK, M - b * HACK);
    const int64_t o = off[b];
    const int w = hack_block_width(off, b, rows);
    const int j0 = min(kseg * segw, w), j1 = min(j0 + segw, w);
    /* col-major: a wavefront reads two columns x 32 rows, each 256 B
     * contiguous; row-major: eight neighbouring columns of a row per 8 lanes */
    const int i = col_major ? (tid & 31) : (tid >> 3);
    const int cl = col_major ? (tid >> 5) : (tid & 7);
    double acc = 0.0;
    if (i < rows) {
        /* slot of column j of row i; the lane's columns are j0 + cl, + 8, ...
         * Eight of them in flight at a time, added in column order (one
         * accumulator): one at a time is two dependent memory round trips
         * per slot, 32 times over -- the launch was latency, not bandwidth */
        const int64_t base = o + (col_major ? i : (int64_t)i * w);
        const int64_t step = col_major ? rows : 1;
        int j = j0 + cl;
        constexpr int U = 8;
        for (; j + 8 * (U - 1) < j1; j += 8 * U) {
            int c[U];
            double v[U], xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t t = base + (int64_t)(j + 8 * u) * step;
                c[u] = ld_stream(ja + t);
                v[u] = ld_stream(as + t);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                xv[u] = x[c[u]];
#pragma unroll
            for (int u = 0; u < U; ++u)
                acc += v[u] * xv[u];
        }
        for (; j < j1; j += 8) {
            const int64_t t = base + (int64_t)j * step;
            acc += ld_stream(as + t) * x[ld_stream(ja + t)];
        }
    }
    red[cl][i] = acc;
    __syncthreads();
    if (tid < HACK) {
        double t = 0.0;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            t += red[c][tid];
        __hip_atomic_store(part + (size_t)g * HACK + tid, t, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int g0 = g - kseg;
    if (tid == 0) /* epoch_arrive (hip_common.h): 1 .. nseg */
        s_seen = (int)epoch_arrive(cnt + g0, epoch);
    __syncthreads();
    if (s_seen != nseg)
        return;
    /* the block's last segment: 8 lanes per row add every 8th segment's
     * partial, then the eight are added in lane order -- a fixed order */
    {
        const int r = tid & 31, c = tid >> 5;
        const double sum = ordered_partial_sum<8>(part + (size_t)g0 * HACK + r,
                                                  nseg, c, 8, HACK);
        __syncthreads(); /* red[] of the column lanes is no longer needed */
        red[c][r] = sum;
    }
    __syncthreads();
    if (tid < rows) {
        double sum = 0.0;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            sum += red[c][tid];
        y[(int64_t)b * HACK + tid] = sum;
    }
    if (tid == 0)
        epoch_rearm(cnt + g0, epoch);
}

/* ------------------------------------------------------------------ */
int hll_launch_kernel(const spmv_hll_dev *H, int kernel, int waves,
                      int variant, const double *x, double *y, int b0, int b1,
                      hipStream_t s) {
    (void)hipGetLastError(); /* an earlier caller's unread error is not ours */
    if (!H || !x || !y || b0 < 0 || b1 > H->nb || b0 > b1)
        return -EINVAL;
    if ((kernel == 0 || kernel == 3) == (H->col_major != 0))
        return -EINVAL; /* layout of the handle does not fit the kernel */
#ifndef SPMV_ABLATIONS
    /* product build: only the documented bits (spmv_engine.h: 0, 1, 2 the
     * workgroup orders; 29 belongs to the timed loops).  The ablation arms
     * (bits 4-6: pipeline depths, an arm whose result is wrong by design) are
     * compiled only with -DSPMV_ABLATIONS (make abl) */
    if (variant & ~(1 | 2 | 4 | SPMV_VARIANT_TIMING_BITS))
        return -EINVAL;
#endif
    if (b0 == b1)
        return 0;
    /* workgroup order: variant bit 0 hardware, bit 1 XCD ranges, bit 2
     * grouped; none: the handle's (0 hardware / 1 ranges / 2 grouped) */
    const int order = (variant & 1) ? 0 : (variant & 2) ? 1 : (variant & 4) ? 2
                                                        : H->order;
    const int threads = waves * WAVE;
    const long long lanes = (long long)(b1 - b0) * HACK;
    const int wide = H->n_wide_seg > 0 ? HLL_WIDE : 0;
    /* XCD ranges of this launch: the handle's slot-balanced table for the
     * whole matrix, an even split for a block sub-range (multi-GPU chunks) */
    xcd_ranges xr = H->xcd_blk;
    if (b0 != 0 || b1 != H->nb)
        for (int k = 0; k <= NUM_XCD; ++k) {
            long long c = b0 + ((long long)(b1 - b0) * k / NUM_XCD + 1) / 2 * 2;
            xr.first[k] = k == NUM_XCD || c > b1 ? b1 : (int)c;
        }
    int xmax = 0; /* longest range, in blocks */
    for (int k = 0; k < NUM_XCD; ++k)
        xmax = xr.first[k + 1] - xr.first[k] > xmax
                   ? xr.first[k + 1] - xr.first[k] : xmax;
    switch (kernel) {
    case 0:
        hipLaunchKernelGGL(k_hll_row_major,
                           dim3((unsigned)((lanes + threads - 1) / threads)),
                           dim3(threads), 0, s, H->M, b0, b1, wide, H->off,
                           H->ja, H->as, x, y);
        break;
    case 1: {
        /* full blocks through LDS; a ragged last block goes direct */
        int full_end = b1;
        if (b1 == H->nb && (H->M % HACK) != 0)
            full_end = b1 - 1;
        if (full_end > b0) {
            int pairs = (full_end - b0 + 1) / 2;
            size_t lds = (size_t)waves * 2 * CH_SLOTS * (sizeof(double) + sizeof(int));
            const int nwg = (pairs + waves - 1) / waves;
            if (order == 1)
                hipLaunchKernelGGL(k_hll_col_lds<1>,
                                   dim3(NUM_XCD * (((xmax + 1) / 2 + waves - 1) /
                                                   waves)),
                                   dim3(threads), lds, s, b0, full_end, wide, xr,
                                   H->off, H->ja, H->as, x, y);
            else if (order == 2)
                hipLaunchKernelGGL(k_hll_col_lds<2>,
                                   dim3((nwg + NUM_XCD * HLL_GROUP - 1) /
                                        (NUM_XCD * HLL_GROUP) * NUM_XCD * HLL_GROUP),
                                   dim3(threads), lds, s, b0, full_end, wide, xr,
                                   H->off, H->ja, H->as, x, y);
            else
                hipLaunchKernelGGL(k_hll_col_lds<0>, dim3(nwg), dim3(threads),
                                   lds, s, b0, full_end, wide, xr, H->off, H->ja,
                                   H->as, x, y);
        }
        if (full_end < b1)
            hipLaunchKernelGGL((k_hll_col_direct<8, 0>), dim3(1), dim3(WAVE), 0, s,
                               H->M, full_end, b1, wide, xr, H->off, H->ja, H->as, x, y);
        break;
    }
    case 2: {
        /* REMAP grid: 8 x (workgroups of the longest XCD range) */
        const unsigned xgrid =
            NUM_XCD * (unsigned)(((long long)xmax * HACK + threads - 1) / threads);
#ifdef SPMV_ABLATIONS /* experiment arms: `make abl` builds them */
        if (variant & 32) { /* tuning: 4 columns per pipeline stage */
            hipLaunchKernelGGL((k_hll_col_direct<4, 1>), dim3(xgrid),
                               dim3(threads), 0, s, H->M, b0, b1, wide, xr, H->off,
                               H->ja, H->as, x, y);
            break;
        }
        if (variant & 64) { /* tuning: 16 columns per pipeline stage */
            hipLaunchKernelGGL((k_hll_col_direct<16, 1>), dim3(xgrid),
                               dim3(threads), 0, s, H->M, b0, b1, wide, xr, H->off,
                               H->ja, H->as, x, y);
            break;
        }
        if (variant & 16) { /* ABL 1: every gather reads x[0..1]: WRONG y */
            hipLaunchKernelGGL((k_hll_col_direct<8, 1, 1>), dim3(xgrid),
                               dim3(threads), 0, s, H->M, b0, b1, wide, xr, H->off,
                               H->ja, H->as, x, y);
            break;
        }
#endif
        const unsigned hwgrid = (unsigned)((lanes + threads - 1) / threads);
        if (order == 1)
            hipLaunchKernelGGL((k_hll_col_direct<8, 1>), dim3(xgrid),
                               dim3(threads), 0, s, H->M, b0, b1, wide, xr, H->off,
                               H->ja, H->as, x, y);
        else if (order == 2)
            hipLaunchKernelGGL((k_hll_col_direct<8, 2>),
                               dim3((hwgrid + NUM_XCD * HLL_GROUP - 1) /
                                    (NUM_XCD * HLL_GROUP) * NUM_XCD * HLL_GROUP),
                               dim3(threads), 0, s, H->M, b0, b1, wide, xr, H->off,
                               H->ja, H->as, x, y);
        else
            hipLaunchKernelGGL((k_hll_col_direct<8, 0>), dim3(hwgrid),
                               dim3(threads), 0, s, H->M, b0, b1, wide, xr, H->off,
                               H->ja, H->as, x, y);
        break;
    }
    case 3:
        hipLaunchKernelGGL(
            k_hll_subwave_row,
            dim3((unsigned)std::min<long long>((lanes * 16 + threads - 1) / threads,
                                               0xFFFFFFFFll / threads)),
            dim3(threads), 0, s, H->M, b0, b1, wide, H->off, H->ja, H->as, x,
            y);
        break;
    default:
        return -EINVAL;
    }
    if (wide > 0)
        hipLaunchKernelGGL(k_hll_wide, dim3(H->n_wide_seg), dim3(256), 0, s,
                           H->M, b0, b1, H->col_major, H->wide_seg, H->off,
                           H->ja, H->as, x, y, H->wide_part, H->wide_cnt,
                           next_launch_epoch(&H->launch_epoch));
    return hip_errno(hipGetLastError());
}

/*
 * csr_kernels.hip -- fp64 CSR SpMV kernels for gfx950 (wave64).
 *
 * Five kernels fill the five slots of the reference's driver table
 * (reference cuda_csr.cu:19-178, main.c:259-263); the designs are new:
 *
 *   0 thread_row   lane per row.
 *   1 wave_row     one 64-lane wavefront per row, lanes stride the row,
 *                  __shfl_down tree over 64 lanes.
 *   2 subwave_row  G lanes per row (G = 2..32), 64/G rows per wavefront;
 *                  a wavefront reads 64 consecutive entries per load, so the
 *                  JA/AS streams stay fully coalesced whatever the row
 *                  length; segmented __shfl_down(width G) reduction with
 *                  every lane taking part (no early exit before a shuffle,
 *                  unlike reference cuda_csr.cu:72-73).
 *   3 block_row    one workgroup per row: wave partials through LDS.
 *   4 stream       nnz-balanced: a workgroup owns consecutive rows holding
 *                  <= STREAM_NNZ entries (table built at upload) and
 *                  fetches them with coalesced loads; ranges of short rows
 *                  are transposed through LDS (lane per row, HLL-like
 *                  gather order, no reduction), ranges with a long row
 *                  stage products and reduce with a lane team; rows longer
 *                  than the budget get a workgroup to themselves.
 *
 * No MFMA: there is no dense contraction.  Bound: HBM streams (12 B/entry)
 * plus the x gathers.  JA/AS are read once -> non-temporal loads, so they
 * do not evict x from L2 / Infinity Cache.
 */
#include <algorithm>
#include "hip_common.h"

template <typename T> __device__ __forceinline__ T ld_stream(const T *p) {
    return __builtin_nontemporal_load(p);
}

/*
 * Segmented reduction inside a wavefront: sum over groups of G consecutive
 * lanes, result in the first lane of each group (what a __shfl_down(width G)
 * tree gives).  Offsets 8,4,2,1 stay inside a 16-lane DPP row and run on the
 * VALU as row_shl moves (no LDS-crossbar ds_bpermute); only the 16- and
 * 32-lane steps use __shfl_down.
 */
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
    const int hi =
        __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(
        double, ((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}

template <int G> __device__ __forceinline__ double group_sum(double v) {
    if (G >= 64)
        v += __shfl_down(v, 32, 64);
    if (G >= 32)
        v += __shfl_down(v, 16, G >= 64 ? 64 : 32);
    if (G >= 16)
        v += dpp_f64<0x108>(v); /* row_shl:8 */
    if (G >= 8)
        v += dpp_f64<0x104>(v); /* row_shl:4 */
    if (G >= 4)
        v += dpp_f64<0x102>(v); /* row_shl:2 */
    if (G >= 2)
        v += dpp_f64<0x101>(v); /* row_shl:1 */
    return v;
}

/* ------------------------------------------------------------------ */
/* `lrow` > 0 (kernels 0-3): the matrix has rows of more than `lrow` (2048:
 * the stream kernel's entry budget, i.e. rows that own a range) entries;
 * they are left to k_csr_long_seg, launched right after on the same stream
 * (one lane / wavefront / workgroup walking a hub row of 10^5 entries was
 * 6-130 ms of an otherwise 0.05-3 ms launch) */
__global__ void k_csr_thread_row(int r0, int r1, int lrow,
                                 const int *__restrict__ irp,
                                 const int *__restrict__ ja,
                                 const double *__restrict__ as,
                                 const double *__restrict__ x,
                                 double *__restrict__ y) {
    int row = r0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= r1)
        return;
    const int b = irp[row], e = irp[row + 1];
    if (lrow > 0 && e - b > lrow)
        return;
    /* the row's entries in order, four loads in flight (hip_common.h) */
    y[row] = strided_dot<1, 4>(ja, as, x, b, e, 0);
}

/* ------------------------------------------------------------------ */
__global__ void k_csr_wave_row(int r0, int r1, int lrow,
                               const int *__restrict__ irp,
                               const int *__restrict__ ja,
                               const double *__restrict__ as,
                               const double *__restrict__ x,
                               double *__restrict__ y) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = threadIdx.x / WAVE;
    const int waves = blockDim.x / WAVE;
    /* grid-stride over the rows (wave-uniform): a launch holds fewer than 2^32
     * work-items, i.e. 2^26 wavefronts -- a matrix at the entry-count limit
     * (67M rows x 32) has more rows than that */
    for (long long row = (long long)r0 + (long long)blockIdx.x * waves + wave;
         row < r1; row += (long long)gridDim.x * waves) {
        const int beg = irp[row], end = irp[row + 1];
        if (lrow > 0 && end - beg > lrow)
            continue; /* k_csr_long_seg's row */
        double acc = strided_dot<WAVE, 4>(ja, as, x, beg, end, lane);
        acc = group_sum<WAVE>(acc);
        if (lane == 0)
            y[row] = acc;
    }
}

/* ------------------------------------------------------------------ */
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    /* contiguous range of the grid per XCD (bijective for any nblk): row
     * tiles that share an x window then meet in the same L2 */
    const int nx = 8;
    int q = nblk / nx, r = nblk % nx;
    int x = bid % nx, k = bid / nx;
    return x * q + (x < r ? x : r) + k;
}

/*
 * G lanes per row, P rows per lane group: a wavefront owns P*(64/G)
 * consecutive rows.  The P passes are independent, so their IRP, JA/AS and x
 * loads are all in flight together (P x 768 B of stream per wavefront
 * instead of 768 B): the kernel is latency-bound otherwise (1 pass 1.30 ms,
 * 4 passes 0.83 ms, 8 passes 0.79 ms on banded 10M x 32).
 */
#define CSR_GROUP 32 /* grouped order: runs of 32 workgroups per XCD */
__device__ __forceinline__ int xcd_grouped(int bid) {
    const int xx = bid % 8, kk = bid / 8;
    return ((kk / CSR_GROUP) * 8 + xx) * CSR_GROUP + kk % CSR_GROUP;
}

/* UNI: every row of the matrix holds exactly `ulen` entries (the handle
 * found that out at upload), so IRP[r] = r * ulen needs no load: the wavefront
 * fetches JA / AS at once instead of one memory latency later.  A launch of
 * this kernel is a few wave lifetimes long on a 1M-row matrix and a lifetime
 * is three dependent latencies (IRP -> JA/AS -> x): config 2 (1M x 16,
 * flushed, 256-lane workgroups) 0.0494 -> 0.0447 ms; banded 10M x 32 0.872 ->
 * 0.859; no change beyond noise on matrices whose launch is long. */
template <int G, int P, int ORDER, bool UNI>
__global__ void k_csr_subwave_row(int r0, int r1, int ulen, int lrow,
                                  const int *__restrict__ irp,
                                  const int *__restrict__ ja,
                                  const double *__restrict__ as,
                                  const double *__restrict__ x,
                                  double *__restrict__ y) {
    constexpr int RPP = WAVE / G; /* rows per pass */
    const int lane = threadIdx.x & (WAVE - 1);
    const int sub = lane & (G - 1);
    /* ORDER 0: hardware order, 1: XCD-contiguous equal ranges, 2: grouped
     * (the grid is then padded to a multiple of 8 x CSR_GROUP; rows beyond
     * r1 are masked below) */
    const int bid = ORDER == 1 ? xcd_remap(blockIdx.x, gridDim.x)
                    : ORDER == 2 ? xcd_grouped(blockIdx.x) : (int)blockIdx.x;
    const long long wave_global =
        ((long long)bid * blockDim.x + threadIdx.x) / WAVE;
    const long long rbase = (long long)r0 + wave_global * (P * RPP) + lane / G;

    int beg[P], end[P];
    bool mine[P]; /* this launch writes the row (not beyond r1, not a long row) */
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const long long row = rbase + p * RPP;
        const bool live = row < r1;
        if (UNI) {
            beg[p] = live ? (int)(row * ulen) : 0;
            end[p] = live ? beg[p] + ulen : 0;
        } else {
            beg[p] = live ? irp[row] : 0;
            end[p] = live ? irp[row + 1] : 0;
        }
        mine[p] = live;
        if (lrow > 0 && end[p] - beg[p] > lrow) { /* k_csr_long_seg's row */
            end[p] = beg[p];
            mine[p] = false;
        }
    }
    int c[P];
    double a[P], acc[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        /* offsets relative to the row's first entry: beg + sub (+ G below)
         * must not be formed in 32 bits next to INT32_MAX */
        const bool has = sub < end[p] - beg[p];
        c[p] = has ? ld_stream(ja + beg[p] + sub) : -1;
        a[p] = has ? ld_stream(as + beg[p] + sub) : 0.0;
    }
#pragma unroll
    for (int p = 0; p < P; ++p)
        acc[p] = c[p] >= 0 ? a[p] * x[c[p]] : 0.0;
#pragma unroll
    for (int p = 0; p < P; ++p)
        for (int k = sub + G, n = end[p] - beg[p]; k < n; k += G)
            acc[p] += ld_stream(as + beg[p] + k) * x[ld_stream(ja + beg[p] + k)];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        acc[p] = group_sum<G>(acc[p]);
        const long long row = rbase + p * RPP;
        if (sub == 0 && mine[p])
            y[row] = acc[p];
    }
}

/* ------------------------------------------------------------------ */
__global__ void k_csr_block_row(int r0, int r1, int lrow,
                                const int *__restrict__ irp,
                                const int *__restrict__ ja,
                                const double *__restrict__ as,
                                const double *__restrict__ x,
                                double *__restrict__ y) {
    __shared__ double part[16];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = threadIdx.x / WAVE;
    const int waves = blockDim.x / WAVE;
    for (int row = r0 + blockIdx.x; row < r1; row += gridDim.x) {
        double acc = 0.0;
        const int beg = irp[row], end = irp[row + 1];
        if (lrow > 0 && end - beg > lrow)
            continue; /* block-uniform; no barrier was entered for this row */
        /* relative to the row: beg + k + blockDim.x must not be formed in 32
         * bits next to INT32_MAX (the entry count's limit) */
        const int *rj = ja + beg;
        const double *ra = as + beg;
        for (int k = threadIdx.x, n = end - beg; k < n; k += blockDim.x)
            acc += ld_stream(ra + k) * x[ld_stream(rj + k)];
#pragma unroll
        for (int d = WAVE / 2; d > 0; d >>= 1)
            acc += __shfl_down(acc, d, WAVE);
        if (lane == 0)
            part[wave] = acc;
        __syncthreads();
        if (wave == 0) {
            double t = lane < waves ? part[lane] : 0.0;
#pragma unroll
            for (int d = 8; d > 0; d >>= 1)
                t += __shfl_down(t, d, 16);
            if (lane == 0)
                y[row] = t;
        }
        __syncthreads();
    }
}

/* ------------------------------------------------------------------ */
/*
 * stream: workgroup k owns rows [rowblk[k], rowblk[k+1]) holding at most
 * STREAM_NNZ entries (table built at upload), or a single longer row.  The
 * range's JA/AS are fetched with coalesced loads; what happens next depends
 * on the rows of the range (mode bit of the table, set at upload):
 *
 *  transposed (every row <= STREAM_ROW_T entries): JA/AS go to LDS, then
 *    lane r walks row r out of LDS and gathers x itself.  At step j the
 *    lanes of a wavefront hold the j-th entry of ADJACENT rows -- the access
 *    order of the col-major HLL kernels, which touches several times fewer
 *    cache lines per gather instruction than lanes-along-the-row on matrices
 *    with row-to-row locality -- and the row sum is serial in the lane, no
 *    cross-lane reduction.  LDS indices are skewed by one slot every 32 so
 *    equal-length rows do not collide on a bank.  (banded 10M x 32: 0.80 ms
 *    vs 1.12 ms for the cooperative form; random W = 2048: 0.89 vs 1.64.)
 *
 *  cooperative (some row is long): the products a_ij * x_j are formed in
 *    load order and staged in LDS; a team of G lanes (G from the mean row
 *    length) sums each row with a segmented reduction, so one long row does
 *    not serialise on a lane.  (rows of 4-8 entries with a 128-entry row in
 *    every 64: 0.51 ms vs 0.72 ms transposed.)
 */
#define TSKEW(k) ((k) + ((k) >> 5))
#define STREAM_LDS (STREAM_NNZ + STREAM_NNZ / 32 + 1)

template <int T>
__device__ __forceinline__ void stream_rows(int tid, int rows, const int *rowptr,
                                            const int *s_ja, const double *s_val,
                                            const double *__restrict__ x,
                                            double *__restrict__ y_range) {
    const int sub = tid % T;
    /* STREAM_THREADS / T lane teams; a range of more rows than that (short
     * rows: up to STREAM_ROWS of them, T = 1) gives every team several rows */
    for (int r = tid / T; r < rows; r += STREAM_THREADS / T) {
    double acc = 0.0;
    {
        const int rz = rowptr[r + 1];
        int k = rowptr[r] + sub;
        for (; k + 3 * T < rz; k += 4 * T) {
            const int c0 = s_ja[TSKEW(k)], c1 = s_ja[TSKEW(k + T)];
            const int c2 = s_ja[TSKEW(k + 2 * T)], c3 = s_ja[TSKEW(k + 3 * T)];
            const double v0 = s_val[TSKEW(k)], v1 = s_val[TSKEW(k + T)];
            const double v2 = s_val[TSKEW(k + 2 * T)];
            const double v3 = s_val[TSKEW(k + 3 * T)];
            const double x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
            acc += v0 * x0;
            acc += v1 * x1;
            acc += v2 * x2;
            acc += v3 * x3;
        }
        for (; k < rz; k += T)
            acc += s_val[TSKEW(k)] * x[s_ja[TSKEW(k)]];
    }
    acc = group_sum<T>(acc); /* a team's lanes run the same trip count */
    if (sub == 0)
        y_range[r] = acc;
    }
}

typedef int s_v4i __attribute__((ext_vector_type(4)));
typedef double s_v2d __attribute__((ext_vector_type(2)));

/* WIDE: ranges in transposed mode fetch JA / AS with 16-byte loads from the
 * 16-byte boundary below the range's first entry (a wavefront instruction
 * covers 1 KiB of whole lines: 6 load instructions per lane instead of 16);
 * the LDS transposition absorbs the shift and the fact that a lane's JA and
 * AS elements are different entries.  The device arrays carry 2056 entries
 * of slack so the last range may read past NZ. */
template <bool WIDE>
__global__ void __launch_bounds__(STREAM_THREADS)
    k_csr_stream(int n_rowblk, int grouped, const int2 *__restrict__ rowblk,
                 const unsigned char *__restrict__ mode,
                 const int *__restrict__ irp, const int *__restrict__ ja,
                 const double *__restrict__ as, const double *__restrict__ x,
                 double *__restrict__ y, double *seg_partial,
                 unsigned long long *seg_count, unsigned epoch) {
    __shared__ double s_val[STREAM_LDS]; /* AS (transposed) or products */
    __shared__ int s_ja[STREAM_LDS];
    __shared__ double part[STREAM_THREADS / WAVE];
    __shared__ int rowptr[STREAM_ROWS + 1]; /* this range's slice of IRP */
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    /* Which range a workgroup runs (workgroups are dealt to the XCDs
     * round-robin): its own index (hardware order) or grouped runs of 32
     * consecutive ranges per XCD, the runs dealt round-robin (+2.5..4.7 % on
     * 10M-row matrices: banded 0.652 vs 0.683 ms, W = 2^11 0.803 vs 0.830,
     * 27-point stencil 0.570 vs 0.584; 1M x 16: 0.0471 vs 0.0462).  XCD-
     * contiguous runs of equal work -- eight distant regions of JA/AS
     * streamed at once -- were measured SLOWER than hardware order (banded
     * 0.738 vs 0.704, stencil 0.621 vs 0.605, 1M x 16 0.049 vs 0.046). */
    const int rb = grouped ? xcd_grouped(blockIdx.x) : (int)blockIdx.x;
    if (rb >= n_rowblk)
        return;
    const int2 t_a = rowblk[rb], t_z = rowblk[rb + 1]; /* (row, entry) */
    const int row_a = t_a.x, row_b = t_z.x;
    const int beg = t_a.y, end = t_z.y;
    const int cnt = end - beg;
    const int rows = row_b - row_a;

    const int md = mode[rb];
    if (md == 2 || cnt > STREAM_NNZ) {
        /* one long row (up to STREAM_LONG_ROW entries), or one SEGMENT of a
         * longer one (mode 2: entries [beg, end) of row row_a, cut at
         * multiples of STREAM_SEG): every lane strides the entries,
         * block-wide reduction */
        double acc = strided_dot<STREAM_THREADS, 8>(ja, as, x, beg, end, tid);
        acc = group_sum<WAVE>(acc);
        if (lane == 0)
            part[tid / WAVE] = acc;
        __syncthreads();
        /* The row's segments are consecutive ranges rb0 .. rb0 + nseg - 1.
         * Each leaves its partial sum; the LAST to arrive (counter at the
         * row's first range) adds them up in a fixed order -- the same
         * whatever the arrival order, so the result is deterministic -- and
         * writes y.  The counter carries the launch's epoch (epoch_arrive,
         * hip_common.h): arrivals a launch that never completed left behind
         * do not count; the last arriver leaves the word the next launch expects.  Agent-scope atomics:
         * the segments run on different XCDs, whose L2s are not coherent for
         * plain loads and stores. */
        __shared__ int s_last;
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < STREAM_THREADS / WAVE; ++w)
                t += part[w];
            s_last = 0;
            if (md != 2) {
                y[row_a] = t;
            } else {
                const int b0 = irp[row_a];
                const int nseg =
                    (irp[row_a + 1] - b0 + STREAM_SEG - 1) / STREAM_SEG;
                const int rb0 = rb - (beg - b0) / STREAM_SEG;
                __hip_atomic_store(seg_partial + rb, t, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                const unsigned seen = epoch_arrive(seg_count + rb0, epoch);
                s_last = seen == (unsigned)nseg ? nseg : 0;
            }
        }
        __syncthreads();
        if (s_last && tid < WAVE) { /* one wavefront adds the partials */
            const int n = s_last;
            const int first = rb - (beg - irp[row_a]) / STREAM_SEG;
            const double sum = wave_ordered_sum(seg_partial + first, n, lane);
            if (lane == 0) {
                y[row_a] = sum;
                epoch_rearm(seg_count + first, epoch);
            }
        }
        return;
    }

    if (WIDE && md == 0 && cnt + (beg & 3) <= STREAM_NNZ) {
        const int d = beg & 3; /* entries between the 16-B boundary and beg */
        const int *ja_al = ja + (beg - d);
        const double *as_al = as + (beg - d);
        s_v4i cj[2];
        s_v2d ca[4];
        for (int r = tid; r < rows; r += STREAM_THREADS)
            rowptr[r] = irp[row_a + r] - beg;
        if (tid == 0)
            rowptr[rows] = cnt;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            cj[i] = ld_stream((const s_v4i *)(ja_al + (i * STREAM_THREADS + tid) * 4));
#pragma unroll
        for (int i = 0; i < 4; ++i)
            ca[i] = ld_stream((const s_v2d *)(as_al + (i * STREAM_THREADS + tid) * 2));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = (i * STREAM_THREADS + tid) * 4 + j - d;
                if (k >= 0 && k < cnt)
                    s_ja[TSKEW(k)] = cj[i][j];
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = (i * STREAM_THREADS + tid) * 2 + j - d;
                if (k >= 0 && k < cnt)
                    s_val[TSKEW(k)] = ca[i][j];
            }
        __syncthreads();
        if (rows * 4 <= STREAM_THREADS)
            stream_rows<4>(tid, rows, rowptr, s_ja, s_val, x, y + row_a);
        else if (rows * 2 <= STREAM_THREADS)
            stream_rows<2>(tid, rows, rowptr, s_ja, s_val, x, y + row_a);
        else
            stream_rows<1>(tid, rows, rowptr, s_ja, s_val, x, y + row_a);
        return;
    }

    /* coalesced fetch of the range's entries (all loads of a lane issued
     * together) and of its row offsets */
    constexpr int E = STREAM_NNZ / STREAM_THREADS;
    int c[E];
    double a[E];
    for (int r = tid; r < rows; r += STREAM_THREADS) /* <= STREAM_ROWS rows */
        rowptr[r] = irp[row_a + r] - beg;
    if (tid == 0)
        rowptr[rows] = cnt;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = tid + e * STREAM_THREADS;
        const bool has = k < cnt;
        c[e] = has ? ld_stream(ja + beg + k) : -1;
        a[e] = has ? ld_stream(as + beg + k) : 0.0;
    }

    if (md == 0) { /* ---- transposed ---- */
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int k = tid + e * STREAM_THREADS;
            if (c[e] >= 0) {
                s_ja[TSKEW(k)] = c[e];
                s_val[TSKEW(k)] = a[e];
            }
        }
        __syncthreads();
        /* T lanes per row so that all 256 lanes work when the range has
         * few rows (64 rows of 32 entries: T = 4); lane `sub` of a team takes
         * entries sub, sub + T, ... -- neighbours in the row, usually the
         * same line of x */
        if (rows * 4 <= STREAM_THREADS)
            stream_rows<4>(tid, rows, rowptr, s_ja, s_val, x, y + row_a);
        else if (rows * 2 <= STREAM_THREADS)
            stream_rows<2>(tid, rows, rowptr, s_ja, s_val, x, y + row_a);
        else
            stream_rows<1>(tid, rows, rowptr, s_ja, s_val, x, y + row_a);
        return;
    }

    /* ---- cooperative: products in load order, lane team per row ---- */
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (c[e] >= 0)
            s_val[tid + e * STREAM_THREADS] = a[e] * x[c[e]];
    __syncthreads();
    int g = 1;
    while (g < WAVE && g * rows * 2 <= cnt)
        g <<= 1; /* g ~ mean length / 2, power of two, <= 64 */
    const int sub = tid & (g - 1);
    const int per_pass = STREAM_THREADS / g;
    for (int r = tid / g; r < rows + (per_pass - rows % per_pass) % per_pass;
         r += per_pass) {
        double acc = 0.0;
        const bool live = r < rows;
        if (live) {
            const int ra = rowptr[r], rz = rowptr[r + 1];
            for (int k = ra + sub; k < rz; k += g)
                acc += s_val[k];
        }
        for (int d = g >> 1; d > 0; d >>= 1)
            acc += __shfl_down(acc, d, WAVE);
        if (live && sub == 0)
            y[row_a + r] = acc;
    }
}

/* ------------------------------------------------------------------ */
/*
 * The long rows of kernels 0-3: workgroup g sums range long_rb[g] of the
 * stream kernel's table -- one whole row of 2049 .. STREAM_LONG_ROW entries,
 * or (mode 2) one STREAM_SEG-entry segment of a longer row -- exactly as
 * k_csr_stream does (same strides, same partial sums, arrival counters and
 * summation order: the two kernels give the same bits for such a row).  Even
 * 2048 entries are 512 dependent steps for a 4-lane team (0.25 ms) -- the
 * power-law matrices have hundreds of such rows.  Rows outside [r0, r1) belong
 * to another launch of a chunked exchange.
 */
__global__ void __launch_bounds__(STREAM_THREADS)
    k_csr_long_seg(int r0, int r1, const int *__restrict__ long_rb,
                   const int2 *__restrict__ rowblk,
                   const unsigned char *__restrict__ mode,
                   const int *__restrict__ irp,
                   const int *__restrict__ ja, const double *__restrict__ as,
                   const double *__restrict__ x, double *__restrict__ y,
                   double *seg_partial, unsigned long long *seg_count,
                   unsigned epoch) {
    __shared__ double part[STREAM_THREADS / WAVE];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int rb = long_rb[blockIdx.x];
    const int2 t_a = rowblk[rb], t_z = rowblk[rb + 1];
    const int row = t_a.x, beg = t_a.y, end = t_z.y;
    if (row < r0 || row >= r1)
        return;
    double acc = strided_dot<STREAM_THREADS, 8>(ja, as, x, beg, end, tid);
    acc = group_sum<WAVE>(acc);
    if (lane == 0)
        part[tid / WAVE] = acc;
    __syncthreads();
    __shared__ int s_last;
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < STREAM_THREADS / WAVE; ++w)
            t += part[w];
        s_last = 0;
        if (mode[rb] != 2) { /* a whole row of 2049 .. STREAM_LONG_ROW entries */
            y[row] = t;
        } else {
            const int b0 = irp[row];
            const int nseg = (irp[row + 1] - b0 + STREAM_SEG - 1) / STREAM_SEG;
            const int rb0 = rb - (beg - b0) / STREAM_SEG;
            __hip_atomic_store(seg_partial + rb, t, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            const unsigned seen = epoch_arrive(seg_count + rb0, epoch);
            s_last = seen == (unsigned)nseg ? nseg : 0;
        }
    }
    __syncthreads();
    if (s_last && tid < WAVE) {
        const int first = rb - (beg - irp[row]) / STREAM_SEG;
        const double sum = wave_ordered_sum(seg_partial + first, s_last, lane);
        if (lane == 0) {
            y[row] = sum;
            epoch_rearm(seg_count + first, epoch);
        }
    }
}

/* ------------------------------------------------------------------ */
static int pick_group(const spmv_csr_dev *A, int group) {
    if (group >= 2 && group <= 32 && (group & (group - 1)) == 0)
        return group;
    double mean = A->M > 0 ? (double)A->NZ / A->M : 1.0;
    int g = 2;
    while (g < 32 && g < mean)
        g <<= 1;
    return g;
}

template <int G, int P, bool UNI>
static void launch_subwave_u(int r0, int r1, int threads, int order,
                             const spmv_csr_dev *A, const double *x, double *y,
                             hipStream_t s) {
    const int rows_per_wave = P * (WAVE / G);
    long long waves = ((long long)(r1 - r0) + rows_per_wave - 1) / rows_per_wave;
    long long wpb = threads / WAVE;
    unsigned grid = (unsigned)((waves + wpb - 1) / wpb);
    const int ulen = A->uniform_len;
    const int lrow = A->n_long_rb > 0 ? STREAM_NNZ : 0;
    if (order == 1)
        hipLaunchKernelGGL((k_csr_subwave_row<G, P, 1, UNI>), dim3(grid),
                           dim3(threads), 0, s, r0, r1, ulen, lrow, A->irp,
                           A->ja, A->as, x, y);
    else if (order == 2)
        hipLaunchKernelGGL((k_csr_subwave_row<G, P, 2, UNI>),
                           dim3((grid + 8 * CSR_GROUP - 1) / (8 * CSR_GROUP) *
                                8 * CSR_GROUP),
                           dim3(threads), 0, s, r0, r1, ulen, lrow, A->irp,
                           A->ja, A->as, x, y);
    else
        hipLaunchKernelGGL((k_csr_subwave_row<G, P, 0, UNI>), dim3(grid),
                           dim3(threads), 0, s, r0, r1, ulen, lrow, A->irp,
                           A->ja, A->as, x, y);
}

/* order bit 8 (0x100) set by the caller: the matrix has a constant row
 * length and the launch may use it */
template <int G, int P>
static void launch_subwave_p(int r0, int r1, int threads, int order,
                             const spmv_csr_dev *A, const double *x, double *y,
                             hipStream_t s) {
    if (order & 0x100)
        launch_subwave_u<G, P, true>(r0, r1, threads, order & 0xff, A, x, y, s);
    else
        launch_subwave_u<G, P, false>(r0, r1, threads, order & 0xff, A, x, y, s);
}

/* passes: independent row groups per wavefront (tuning, variant bits 2-3;
 * carried per launch -- the launch path keeps no process-global state) */
template <int G>
static void launch_subwave(int passes, int r0, int r1, int threads, int remap,
                           const spmv_csr_dev *A, const double *x, double *y,
                           hipStream_t s) {
    switch (passes) {
    case 2:
        launch_subwave_p<G, 2>(r0, r1, threads, remap, A, x, y, s);
        break;
    case 4:
        launch_subwave_p<G, 4>(r0, r1, threads, remap, A, x, y, s);
        break;
    default: /* 8 passes: 0.79 ms vs 0.83 ms (4) on banded 10M x 32 */
        launch_subwave_p<G, 8>(r0, r1, threads, remap, A, x, y, s);
        break;
    }
}

int csr_launch_kernel(const spmv_csr_dev *A, int kernel, int waves, int group,
                      int variant, const double *x, double *y, int r0, int r1,
                      hipStream_t s) {
    (void)hipGetLastError(); /* an earlier caller's unread error is not ours */
    if (!A || !x || !y || r0 < 0 || r1 > A->M || r0 > r1)
        return -EINVAL;
#ifdef SPMV_ABLATIONS /* passes of the sub-wave kernel: bits 2-3 (make abl) */
    const int passes = (variant & 4) ? 4 : (variant & 8) ? 2 : 8;
#else
    /* product build: the documented bits only (spmv_engine.h: 0, 1, 5 the
     * sub-wave kernel's workgroup order; 4, 5, 6 the stream kernel's loads
     * and order; 9 keeps the IRP loads) */
    if (variant & ~(1 | 2 | 16 | 32 | 64 | 512 | SPMV_VARIANT_TIMING_BITS))
        return -EINVAL;
    const int passes = 8;
#endif
    /* workgroup order of the sub-wave kernel: variant bit 0 hardware, bit 1
     * XCD-contiguous ranges, bit 5 grouped runs; none: the handle's
     * (0 / 1 / 2, spmv_csr_autotune) */
    int remap = (variant & 1) ? 0 : (variant & 2) ? 1 : (variant & 32) ? 2
                                                  : A->order;
    /* constant row length: the sub-wave kernel skips the IRP loads (variant
     * bit 9 keeps them, for A/B and for the tests of the general path) */
    if (A->uniform_len > 0 && !(variant & 512))
        remap |= 0x100;
    if (r0 == r1)
        return 0;
    /* this launch's number for the last-arriver counters of the long rows */
    const unsigned epoch =
        A->seg_count ? next_launch_epoch(&A->launch_epoch) : 0u;
    const int threads = waves * WAVE;
    const int rows = r1 - r0;
    /* rows beyond STREAM_NNZ entries (they own a range of the stream table):
     * kernels 0-3 skip them, a second launch sums them (k_csr_long_seg) */
    const int lrow = A->n_long_rb > 0 ? STREAM_NNZ : 0;
    /* the stream kernel's row-block table covers the whole matrix; a row
     * sub-range (chunked multi-GPU overlap) runs the sub-wave kernel */
    if (kernel == 4 && (r0 != 0 || r1 != A->M))
        kernel = 2;
    switch (kernel) {
    case 0:
        hipLaunchKernelGGL(k_csr_thread_row,
                           dim3((rows + threads - 1) / threads), dim3(threads),
                           0, s, r0, r1, lrow, A->irp, A->ja, A->as, x, y);
        break;
    case 1:
        hipLaunchKernelGGL(k_csr_wave_row,
                           dim3(std::min((rows + waves - 1) / waves,
                                         (int)(0xFFFFFFFFu / (unsigned)threads))),
                           dim3(threads), 0, s, r0, r1, lrow, A->irp, A->ja,
                           A->as, x, y);
        break;
    case 2:
        switch (pick_group(A, group)) {
        case 2:
            launch_subwave<2>(passes, r0, r1, threads, remap, A, x, y, s);
            break;
        case 4:
            launch_subwave<4>(passes, r0, r1, threads, remap, A, x, y, s);
            break;
        case 8:
            launch_subwave<8>(passes, r0, r1, threads, remap, A, x, y, s);
            break;
        case 16:
            launch_subwave<16>(passes, r0, r1, threads, remap, A, x, y, s);
            break;
        default:
            launch_subwave<32>(passes, r0, r1, threads, remap, A, x, y, s);
            break;
        }
        break;
    case 3: {
        int grid = rows < 65536 * 16 ? rows : 65536 * 16;
        hipLaunchKernelGGL(k_csr_block_row, dim3(grid), dim3(threads), 0, s,
                           r0, r1, lrow, A->irp, A->ja, A->as, x, y);
        break;
    }
    case 4: {
        /* Tried in round 2 and dropped on measurement: a persistent,
         * ticket-scheduled, software-pipelined form (next range's JA/AS
         * loads issued under the current range's gathers).  2.5x slower
         * (1M x 16: 0.116 vs 0.046 ms; 10M x 32: 1.87 vs 0.65 ms), with
         * __syncthreads and with LDS-only barriers alike: vector-memory
         * results return in order, so the gathers wait behind the prefetched
         * stream loads they were meant to overlap. */
        /* ... and at WAVEFRONT granularity (a wavefront owns <= 512 entries,
         * four decoupled wavefronts per workgroup, wave-level ordering only,
         * like the thread-per-row HLL kernel): slower everywhere -- 1M x 16
         * 0.0454 vs 0.0438 ms, banded 10M x 32 0.766 vs 0.725, 27-point
         * stencil 0.687 vs 0.608, random W = 2048 0.865 vs 0.798. */
        /* ... and (round 3) a finer form, 128 lanes / 1024 entries per range
         * with its own table for matrices under 2M rows (the thread-per-row
         * HLL kernel gains 10 % from 256- instead of 512-lane workgroups on
         * 1M x 16): no gain there (0.0410 vs 0.0411 ms, read-only flush) and
         * slower elsewhere (random 1M x 32, W = 2048: 0.0932 vs 0.0842;
         * 27-point stencil 1.5M: 0.1153 vs 0.1078). */
        /* ... and (round 3) a wavefront-PIPELINED form built on the lesson of
         * the first attempt: a wavefront walks mini-ranges of <= 508 entries
         * (their own table), transposes each through 6 KiB of LDS and issues
         * the NEXT mini-range's JA / AS / IRP loads right after its last
         * gathers (younger loads, so the gathers' data returns first;
         * straight-line, countable vmcnt), mini-ranges dealt grid-stride so
         * that the chip keeps one compact front.  Correct (parity suite) and
         * the pipelining does what it should -- 1.20 ms without it, 0.76 with
         * 7-8 mini-ranges per wavefront on banded 10M x 32 -- but the
         * per-mini-range work of a lone wavefront (masked LDS transposition,
         * lane-team bookkeeping, cross-lane reductions, ~10 LDS round trips in
         * series) costs more than the workgroup form's idle phases: 0.76 vs
         * 0.70 ms there, 0.0515 vs 0.0414 ms on 1M x 16.  Not kept.  The
         * same pipeline at WORKGROUP level (persistent 256-lane workgroups,
         * grid-stride ranges, next range's loads in registers behind the
         * last gathers, batch count per range precomputed on the host):
         * correct, and slower again -- banded 10M x 32 0.86-0.88 vs 0.71 ms
         * with 3-12 workgroups per CU, 1M x 16 0.0546 vs 0.0421, random
         * W = 2048 0.970 vs 0.755, 27-point stencil 0.733 vs 0.564.  Six
         * short-lived one-shot workgroups per CU, replaced by the hardware
         * as they finish, are the better pipeline. */
        if (A->n_rowblk <= 0)
            break;
        {
            /* ranges in grouped runs of 32 per XCD (variant bit 5) or in
             * hardware order (bit 6); none: the handle's */
            const int grp = (variant & 32) ? 1 : (variant & 64) ? 0
                                                               : A->stream_grouped;
            const unsigned grid =
                grp ? (unsigned)((A->n_rowblk + 8 * CSR_GROUP - 1) /
                                 (8 * CSR_GROUP) * 8 * CSR_GROUP)
                    : (unsigned)A->n_rowblk;
            if (variant & 16) /* tuning: 4- / 8-byte loads only */
                hipLaunchKernelGGL(k_csr_stream<false>, dim3(grid),
                                   dim3(STREAM_THREADS), 0, s, A->n_rowblk, grp,
                                   (const int2 *)A->rowblk, A->rowblk_mode,
                                   A->irp, A->ja, A->as, x, y, A->seg_partial,
                                   A->seg_count, epoch);
            else
                hipLaunchKernelGGL(k_csr_stream<true>, dim3(grid),
                                   dim3(STREAM_THREADS), 0, s, A->n_rowblk, grp,
                                   (const int2 *)A->rowblk, A->rowblk_mode,
                                   A->irp, A->ja, A->as, x, y, A->seg_partial,
                                   A->seg_count, epoch);
        }
        break;
    }
    default:
        return -EINVAL;
    }
    if (kernel != 4 && lrow > 0)
        hipLaunchKernelGGL(k_csr_long_seg, dim3(A->n_long_rb),
                           dim3(STREAM_THREADS), 0, s, r0, r1, A->long_rb,
                           (const int2 *)A->rowblk, A->rowblk_mode, A->irp,
                           A->ja, A->as, x, y, A->seg_partial, A->seg_count,
                           epoch);
    return hip_errno(hipGetLastError());
}

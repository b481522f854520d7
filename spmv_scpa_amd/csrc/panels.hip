/*
 * panels.hip -- column-panel path: keeps the x gathers L2-resident for
 * matrices whose rows reach far (no reference counterpart).
 *
 * Why: on MI355X an 8-byte gather of x that misses L1 moves a whole 128-byte
 * line.  Served by the XCD's 4 MiB L2 that costs L2 bandwidth (~265 G
 * gathers/s); served from beyond L2 (Infinity Cache or HBM) it is bound by
 * the fabric at ~55 G gathers/s (profiles/r01_microbench_mi355x.txt).  A
 * 10M x 10M matrix with columns anywhere therefore spends 5.8 ms on gathers
 * alone, whatever the kernel.
 *
 * What: at upload the entries are re-ordered by (column panel, row), a panel
 * being 2^18 columns = 2 MiB of x.  One launch per panel streams that
 * panel's entries (row, col, value: 16 B each, fully coalesced); every CU of
 * every XCD gathers from the same 2 MiB of x, which stays in L2.  A
 * wavefront owns a contiguous run of entries; products are combined with a
 * segmented scan over the (sorted) row ids and added into y.  A row's run
 * that lies wholly inside one wavefront is updated with a plain
 * read-modify-write (nobody else touches that y element during this launch);
 * runs cut by a wavefront boundary use a hardware fp64 atomic add.  Launches
 * of successive panels are ordered by the stream, y is zeroed first.
 *
 * Cost: 16 B/entry instead of 12, plus 16 B per (row, panel) run for y.
 */
#include <hipcub/hipcub.hpp>

#include "hip_common.h"

struct spmv_panels {
    int shift;      /* log2(columns per panel) */
    int count;      /* panels */
    int64_t nnz;    /* entries kept */
    int *row;       /* [nnz] sorted by (panel, row) */
    int *col;       /* [nnz] */
    double *val;    /* [nnz] */
    int64_t *ptr;   /* HOST [count+1] entry range of each panel */
};

void panels_free(spmv_panels *p) {
    if (!p)
        return;
    (void)hipFree(p->row);
    (void)hipFree(p->col);
    (void)hipFree(p->val);
    free(p->ptr);
    free(p);
}

/* ---- key generation: key = panel * Mpad + row, UINT64_MAX = dropped ---- */
__global__ void k_keys_from_csr(int M, uint64_t Mpad, int shift,
                                const int *__restrict__ irp,
                                const int *__restrict__ ja, uint64_t *key,
                                unsigned *idx) {
    /* G = 8 lanes per row keep the writes reasonably coalesced */
    const int sub = threadIdx.x & 7;
    long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    if (row >= M)
        return;
    for (int k = irp[row] + sub, e = irp[row + 1]; k < e; k += 8) {
        key[k] = (uint64_t)(ja[k] >> shift) * Mpad + (uint64_t)row;
        idx[k] = (unsigned)k;
    }
}

__global__ void k_keys_from_hll(int M, uint64_t Mpad, int shift, int col_major,
                                const int64_t *__restrict__ off,
                                const int *__restrict__ ja,
                                const double *__restrict__ as, uint64_t *key,
                                unsigned *idx) {
    int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= M)
        return;
    int b = row >> 5, i = row & 31;
    int rows = min(32, M - b * 32);
    int64_t o = off[b];
    int w = (int)((unsigned)(off[b + 1] - o) / (unsigned)rows);
    for (int j = 0; j < w; ++j) {
        int64_t t = o + (col_major ? (int64_t)j * rows + i : (int64_t)i * w + j);
        /* pads carry the value 0.0 (hip_hll.h); a slot that is exactly zero
         * contributes nothing and is dropped */
        bool keep = as[t] != 0.0;
        key[t] = keep ? (uint64_t)(ja[t] >> shift) * Mpad + (uint64_t)row
                      : ~(uint64_t)0;
        idx[t] = (unsigned)t;
    }
}

__global__ void k_panel_gather(int64_t n, uint64_t Mpad,
                               const uint64_t *__restrict__ key,
                               const unsigned *__restrict__ idx,
                               const int *__restrict__ ja,
                               const double *__restrict__ as, int *prow,
                               int *pcol, double *pval) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    unsigned t = idx[k];
    prow[k] = (int)(key[k] % Mpad);
    pcol[k] = ja[t];
    pval[k] = as[t];
}

/* first position whose key >= bound[p], one thread per panel boundary */
__global__ void k_lower_bounds(int count, int64_t n, uint64_t Mpad,
                               const uint64_t *__restrict__ key, int64_t *ptr) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > count)
        return;
    uint64_t bound = (uint64_t)p * Mpad;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (key[mid] < bound)
            lo = mid + 1;
        else
            hi = mid;
    }
    ptr[p] = lo;
}

/*
 * Build the panel arrays from `slots` source positions (CSR entries or HLL
 * slots).  ja/as are the source arrays the sorted index points into.
 */
static int panels_build(int M, int N, int64_t slots, int panel_cols,
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
    const int count = (int)((((int64_t)N - 1) >> shift) + 1);
    const uint64_t Mpad = (uint64_t)(M > 0 ? M : 1);
    spmv_panels *P = (spmv_panels *)calloc(1, sizeof *P);
    if (!P)
        return -ENOMEM;
    P->shift = shift;
    P->count = count;
    uint64_t *key[2] = {NULL, NULL};
    unsigned *idx[2] = {NULL, NULL};
    void *tmp = NULL;
    size_t tmp_bytes = 0;
    int64_t *d_ptr = NULL;
    const size_t n = (size_t)(slots > 0 ? slots : 1);
    uint64_t *skey = NULL; /* sorted keys / indices (whichever buffer) */
    unsigned *sidx = NULL;

    for (int k = 0; k < 2; ++k) {
        HIP_TRY(hipMalloc((void **)&key[k], n * sizeof(uint64_t)));
        HIP_TRY(hipMalloc((void **)&idx[k], n * sizeof(unsigned)));
    }
    if (slots > 0) {
        if (irp_or_null)
            hipLaunchKernelGGL(k_keys_from_csr,
                               dim3((unsigned)(((long long)M * 8 + 255) / 256)),
                               dim3(256), 0, 0, M, Mpad, shift, irp_or_null, ja,
                               key[0], idx[0]);
        else
            hipLaunchKernelGGL(k_keys_from_hll, dim3((M + 255) / 256),
                               dim3(256), 0, 0, M, Mpad, shift, col_major,
                               off_or_null, ja, as, key[0], idx[0]);
        HIP_TRY(hipGetLastError());
        {
            hipcub::DoubleBuffer<uint64_t> dk(key[0], key[1]);
            hipcub::DoubleBuffer<unsigned> dv(idx[0], idx[1]);
            /* CSR sources have no dropped slots: sort only the used bits */
            int end_bit = 64;
            if (irp_or_null) {
                end_bit = 1;
                while (end_bit < 64 && (((uint64_t)count * Mpad) >> end_bit))
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
    if (!skey) {
        skey = key[0];
        sidx = idx[0];
    }
    HIP_TRY(hipMalloc((void **)&d_ptr, ((size_t)count + 1) * sizeof(int64_t)));
    hipLaunchKernelGGL(k_lower_bounds, dim3((count + 256) / 256), dim3(256), 0,
                       0, count, slots, Mpad, skey, d_ptr);
    HIP_TRY(hipGetLastError());
    P->ptr = (int64_t *)malloc(((size_t)count + 1) * sizeof(int64_t));
    if (!P->ptr) {
        rc = -ENOMEM;
        goto fail;
    }
    HIP_TRY(hipMemcpy(P->ptr, d_ptr, ((size_t)count + 1) * sizeof(int64_t),
                      hipMemcpyDeviceToHost));
    P->nnz = P->ptr[count]; /* dropped slots sort behind the last panel */
    {
        const size_t m = (size_t)(P->nnz > 0 ? P->nnz : 1);
        HIP_TRY(hipMalloc((void **)&P->row, m * sizeof(int)));
        HIP_TRY(hipMalloc((void **)&P->col, m * sizeof(int)));
        HIP_TRY(hipMalloc((void **)&P->val, m * sizeof(double)));
    }
    if (P->nnz > 0) {
        hipLaunchKernelGGL(k_panel_gather,
                           dim3((unsigned)((P->nnz + 255) / 256)), dim3(256), 0,
                           0, P->nnz, Mpad, skey, sidx, ja, as, P->row,
                           P->col, P->val);
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
    (void)hipFree(d_ptr);
    panels_free(P);
    return rc;
}

/* ------------------------------------------------------------------ */
/* the kernel: one launch per panel                                     */
/* ------------------------------------------------------------------ */
#define PANEL_CHUNKS 8 /* 64-entry chunks per wavefront */

__device__ __forceinline__ double readlane_f64(double v, int l) {
    long long b = __builtin_bit_cast(long long, v);
    int lo = __builtin_amdgcn_readlane((int)b, l);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __builtin_bit_cast(double,
                              ((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}

/*
 * All PANEL_CHUNKS chunks of a wavefront are loaded and gathered up front
 * (8 x 16 B per lane of stream + 8 gathers in flight), then scanned one
 * after the other (the carry between chunks is a register), and the y
 * updates of the whole range are issued as one batch at the end: the only
 * memory round trips on the critical path are stream -> gather -> y.
 */
template <int ABL> /* ablation bits (timing experiments only): 1 = no y
                      update, 2 = no scan, 4 = no x gather, 8 = y via nt */
__global__ void k_panel_spmv(int64_t e0, int64_t e1,
                             const int *__restrict__ prow,
                             const int *__restrict__ pcol,
                             const double *__restrict__ pval,
                             const double *__restrict__ x,
                             double *__restrict__ y) {
    constexpr int E = PANEL_CHUNKS;
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t wave =
        ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int64_t base = e0 + wave * (WAVE * E);
    if (base >= e1)
        return; /* wave-uniform */
    const int64_t end = min(e1, base + (int64_t)WAVE * E);
    /* rows just outside the wavefront's range (same panel), for run edges */
    const int row_before = base > e0 ? prow[base - 1] : -1;
    const int row_after = end < e1 ? prow[end] : -2;

    int r[E];
    double p[E];
    {
        int c[E];
        double v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int64_t k = base + e * WAVE + lane;
            const bool ok = k < end;
            r[e] = ok ? __builtin_nontemporal_load(prow + k) : -3;
            c[e] = ok ? __builtin_nontemporal_load(pcol + k) : 0;
            v[e] = ok ? __builtin_nontemporal_load(pval + k) : 0.0;
        }
#pragma unroll
        for (int e = 0; e < E; ++e)
            p[e] = r[e] >= 0 ? v[e] * ((ABL & 4) ? (double)c[e] : x[c[e]]) : 0.0;
    }

    const unsigned long long below =
        lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    unsigned flush = 0, plain = 0; /* per chunk: y update needed / exclusive */
    int last_row = row_before;
    bool open_inside = false; /* the run crossing a chunk edge began inside */
    double carry = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int64_t k0 = base + e * WAVE;
        if (k0 < end) { /* wave-uniform */
            const int nlast = (int)min((int64_t)WAVE, end - k0) - 1;
            const bool valid = lane <= nlast;
            const bool more = k0 + WAVE < end; /* another chunk follows */
            int r_prev = __shfl_up(r[e], 1, WAVE);
            if (lane == 0)
                r_prev = last_row;
            int r_next = __shfl_down(r[e], 1, WAVE);
            const int next_first =
                (e + 1 < E) ? __builtin_amdgcn_readfirstlane(r[e + 1 < E ? e + 1 : e])
                            : row_after;
            if (lane == nlast)
                r_next = more ? next_first : row_after;
            const bool head = valid && r[e] != r_prev;
            const bool tail = valid && r[e] != r_next;
            if (lane == 0 && valid && !head)
                p[e] += carry; /* run continues from the previous chunk */

            const unsigned long long m = __ballot(head) & below;
            const int segstart = m ? 63 - __clzll((long long)m) : 0;
            const bool started_before = (m == 0ull);
            if (!(ABL & 2)) {
#pragma unroll
                for (int d = 1; d < WAVE; d <<= 1) {
                    const double t = __shfl_up(p[e], d, WAVE);
                    if (lane - d >= segstart)
                        p[e] += t;
                }
            }
            const bool inside = started_before ? open_inside : true;
            /* a run that leaves the wavefront's range is flushed (atomically)
             * by its last lane in the range */
            const bool partial = valid && !tail && lane == nlast && !more;
            if (tail || partial)
                flush |= 1u << e;
            if (tail && inside)
                plain |= 1u << e;

            const bool tail_last = __builtin_amdgcn_readlane((int)tail, nlast);
            const bool sb_last =
                __builtin_amdgcn_readlane((int)started_before, nlast);
            carry = tail_last ? 0.0 : readlane_f64(p[e], nlast);
            open_inside = tail_last ? false : (sb_last ? open_inside : true);
            last_row = __builtin_amdgcn_readlane(r[e], nlast);
        }
    }

    if (ABL & 1) {
        double t = 0;
#pragma unroll
        for (int e = 0; e < E; ++e)
            t += p[e] + flush + plain;
        if (t == 1.2345e300)
            y[0] = t;
        return;
    }
    /* y updates of the whole range in one batch */
    double yo[E];
#pragma unroll
    for (int e = 0; e < E; ++e)
        yo[e] = (plain >> e) & 1 ? y[r[e]] : 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if ((plain >> e) & 1)
            y[r[e]] = yo[e] + p[e]; /* sole owner of this row in this launch */
        else if ((flush >> e) & 1)
            unsafeAtomicAdd(y + r[e], p[e]);
    }
}

int panels_launch(const spmv_panels *P, int M, int waves, int variant,
                  const double *x, double *y, hipStream_t s) {
    if (!P)
        return -EINVAL;
    hipError_t e = hipMemsetAsync(y, 0, (size_t)M * sizeof(double), s);
    if (e != hipSuccess)
        return hip_errno(e);
    const int threads = waves * WAVE;
    const int64_t per_block = (int64_t)waves * WAVE * PANEL_CHUNKS;
    for (int p = 0; p < P->count; ++p) {
        const int64_t e0 = P->ptr[p], e1 = P->ptr[p + 1];
        if (e1 <= e0)
            continue;
        const unsigned grid = (unsigned)((e1 - e0 + per_block - 1) / per_block);
#define PL(A) hipLaunchKernelGGL(k_panel_spmv<A>, dim3(grid), dim3(threads), 0, \
                              s, e0, e1, P->row, P->col, P->val, x, y)
        switch ((variant >> 4) & 7) {
        case 0: PL(0); break;
        case 1: PL(1); break;
        case 2: PL(2); break;
        case 3: PL(3); break;
        case 4: PL(4); break;
        case 5: PL(5); break;
        case 6: PL(6); break;
        default: PL(7); break;
        }
#undef PL
    }
    return hip_errno(hipGetLastError());
}

int panels_from_csr(const spmv_csr_dev *A, int panel_cols, spmv_panels **out) {
    return panels_build(A->M, A->N, A->NZ, panel_cols, A->irp, NULL, 0, A->ja,
                        A->as, out);
}

int panels_from_hll(const spmv_hll_dev *H, int panel_cols, spmv_panels **out) {
    return panels_build(H->M, H->N, H->slots, panel_cols, NULL, H->off,
                        H->col_major, H->ja, H->as, out);
}

int64_t panels_nnz(const spmv_panels *P) { return P ? P->nnz : 0; }
int panels_count(const spmv_panels *P) { return P ? P->count : 0; }

/*
 * engine.hip -- device management, matrix upload / generation / conversion,
 * launch dispatch, event timing and the one-shot C-ABI entry points
 * (API: include/spmv_engine.h, hip_csr.h, hip_hll.h).
 *
 * Everything here is host code around the kernels of csr_kernels.hip and
 * hll_kernels.hip.  There is no CPU fallback: without a usable GPU every
 * entry point returns -ENODEV.
 */
#include <algorithm>
#include <functional>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <thread>
#include <time.h>
#include <unordered_map>
#include <vector>

#include "err.h"
#include "hip_common.h"
#include "hip_csr.h"
#include "hip_hll.h"
#include "spmv_synth.h"
#include "stream_table.h"
#include "tune_blocked.h"
#include "utils.h"

/* Process defaults behind set_*_waves_per_block (the reference's seam sets
 * them before every call).  0 = never set: the launch then picks by size --
 * 512-lane workgroups measured best on 10M-row matrices, 256-lane ones on
 * launches that last only a few workgroup lifetimes (1M x 16, flushed:
 * thread-per-row HLL 0.0383 vs 0.0435 ms, sub-wave CSR 0.0447 vs 0.0461; a
 * pure 192 MB stream likewise prefers many small workgroups,
 * profiles/r03_microbench_small.txt). */
int g_csr_waves = 0;
int g_hll_waves = 0;

static int default_waves(int user_default, long long rows) {
    return user_default > 0 ? user_default : rows < 2000000 ? 4 : 8;
}

/*
 * Live handles.  spmv_csr_release / spmv_hll_release free host memory
 * (free(d), the blocked copy's descriptor): a binding that released a handle
 * twice -- an explicit release() followed by a finaliser, two wrappers around
 * one handle, a finaliser running after an atexit sweep -- would corrupt the
 * host heap.  Every handle is therefore entered here when it is created,
 * together with a process-wide GENERATION number (1, 2, 3 ... never reused),
 * and
 *   - every public entry point that takes a handle checks it first (-EBADF for
 *     a pointer that is not, or no longer, a live handle -- before anything is
 *     allocated or dereferenced);
 *   - a release of anything that is not in the set is ignored AND COUNTED
 *     (spmv_ignored_releases(); one line on stderr per ignored release after
 *     spmv_set_debug(1)), so a double release stays visible;
 *   - spmv_*_release_checked(h, generation) releases only when the live handle
 *     at that address still carries the generation the caller was given at
 *     creation (spmv_handle_generation): a stale wrapper cannot release a NEW
 *     handle that calloc happened to place at the old address.  Bindings with
 *     finalisers (the Python one) use this form.
 * The set and its mutex are heap objects that are never destroyed, so a
 * release that arrives during static destruction still finds them.
 */
struct live_set {
    std::mutex mu;
    std::unordered_map<const void *, uint64_t> handles; /* -> generation */
    uint64_t next_gen = 1;
    long ignored = 0;
    int debug = 0;
};
static live_set &live(void) {
    static live_set *s = new live_set();
    return *s;
}
static void live_add(const void *h) {
    std::lock_guard<std::mutex> g(live().mu);
    live().handles[h] = live().next_gen++;
}
static void live_ignored(const void *h, const char *why) {
    /* caller holds the mutex */
    ++live().ignored;
    if (live().debug)
        fprintf(stderr, "spmv_scpa_amd: release of %p ignored (%s)\n", h, why);
}
/* true exactly once per handle: the caller then owns the teardown.  gen != 0:
 * only when the live handle at this address has that generation */
static bool live_take(const void *h, uint64_t gen = 0) {
    std::lock_guard<std::mutex> g(live().mu);
    auto it = live().handles.find(h);
    if (it == live().handles.end()) {
        live_ignored(h, "not a live handle: released before, or never one");
        return false;
    }
    if (gen && it->second != gen) {
        live_ignored(h, "stale generation: the address now holds a newer handle");
        return false;
    }
    live().handles.erase(it);
    return true;
}
static bool live_has(const void *h) {
    std::lock_guard<std::mutex> g(live().mu);
    return live().handles.count(h) == 1;
}
/* first statement of every public entry point that takes a handle */
#define HANDLE_OK(h)                                                          \
    do {                                                                      \
        if (!(h))                                                             \
            return -EINVAL;                                                   \
        if (!live_has(h))                                                     \
            return -EBADF; /* released, or never a handle */                  \
    } while (0)

/* the blocked path as a candidate: tune_blocked.h with panels.hip's
 * operations (the same template runs under ASan with mock copies) */
struct panels_ops {
    static void free(spmv_panels *p) { panels_free(p); }
    static void set_chain(spmv_panels *p, int v) { panels_set_chain(p, v); }
    static void set_waves(spmv_panels *p, int v) { panels_set_waves(p, v); }
    static void set_order(spmv_panels *p, int v) { panels_set_order(p, v); }
    static int balanced_tile_rows(int M, int max_rows) {
        return panels_balanced_tile_rows(M, max_rows);
    }
    static const char *build_phases(void) { return panels_last_build_phases(); }
    static double now_s(void) {
        struct timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
    }
};

/* what the selector did, kept in the handle (spmv_*_tune_log) */
struct tune_log {
    std::string text;
    void operator()(const char *line) {
        text += line;
        text += '\n';
    }
    char *release(void) const { return strdup(text.c_str()); }
};

extern "C" {

/* 0.3: spmv_panel_opts.bucket_order; 0.4: spmv_panel_opts.struct_size (first
 * field), spmv_*_release_checked, handle checks on every entry point */
const char *spmv_version(void) { return "spmv_scpa_amd 0.6 gfx950"; }

/* HIP_VERSION of the headers this library was compiled against, and of the
 * runtime it is bound to now (hipRuntimeGetVersion; needs no device).  The
 * binding compares them: a process that maps torch's bundled runtime runs
 * this library on THAT build of HIP (__init__.py). */
int spmv_hip_build_version(void) { return HIP_VERSION; }
int spmv_hip_runtime_version(void) {
    int v = 0;
    return hipRuntimeGetVersion(&v) == hipSuccess ? v : -EIO;
}

/* "product", or "ablations" for a -DSPMV_ABLATIONS build (make abl): only
 * that flavour understands the experiment bits of spmv_launch_opts.variant */
const char *spmv_build_flavour(void) {
#ifdef SPMV_ABLATIONS
    return "ablations";
#else
    return "product";
#endif
}

int spmv_live_handles(void) {
    std::lock_guard<std::mutex> g(live().mu);
    return (int)live().handles.size();
}

long spmv_ignored_releases(void) {
    std::lock_guard<std::mutex> g(live().mu);
    return live().ignored;
}

void spmv_set_debug(int on) {
    std::lock_guard<std::mutex> g(live().mu);
    live().debug = on ? 1 : 0;
}

uint64_t spmv_handle_generation(const void *handle) {
    std::lock_guard<std::mutex> g(live().mu);
    auto it = live().handles.find(handle);
    return it == live().handles.end() ? 0 : it->second;
}

/*
 * TEST HOOK: overwrite every last-arriver counter of the handle (the long
 * rows of the CSR kernels / the wide hack blocks of the HLL kernels / the long
 * rows beside a blocked copy) with what an INCOMPLETE launch leaves behind:
 * a foreign launch number and a count.  The next launches must not care
 * (epoch_arrive, hip_common.h).  Returns the counters touched, < 0 on error.
 */
int spmv_csr_debug_stale_arrivals(spmv_csr_dev *A) {
    HANDLE_OK(A);
    int n = 0;
    if (A->seg_count && A->n_rowblk > 0) {
        HIP_RET(hipMemset(A->seg_count, 0x01,
                          (size_t)A->n_rowblk * sizeof(unsigned long long)));
        n += A->n_rowblk;
    }
    const int p = panels_debug_stale_arrivals(A->panels);
    return p < 0 ? p : n + p;
}

int spmv_hll_debug_stale_arrivals(spmv_hll_dev *H) {
    HANDLE_OK(H);
    int n = 0;
    if (H->wide_cnt && H->n_wide_seg > 0) {
        HIP_RET(hipMemset(H->wide_cnt, 0x01,
                          (size_t)H->n_wide_seg * sizeof(unsigned long long)));
        n += H->n_wide_seg;
    }
    const int p = panels_debug_stale_arrivals(H->panels);
    return p < 0 ? p : n + p;
}

/* 1..16; 0 (or less) returns to the size-based default above */
void set_csr_waves_per_block(int waves) {
    g_csr_waves = waves < 1 ? 0 : (waves > 16 ? 16 : waves);
}

void set_hll_waves_per_block(int waves) {
    g_hll_waves = waves < 1 ? 0 : (waves > 16 ? 16 : waves);
}

/* ------------------------------------------------------------------ */
/* devices and raw memory                                               */
/* ------------------------------------------------------------------ */

int spmv_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int spmv_set_device(int device) {
    if (spmv_device_count() == 0)
        return -ENODEV;
    HIP_RET(hipSetDevice(device));
    return 0;
}

int spmv_get_device(void) {
    int d = 0;
    if (spmv_device_count() == 0)
        return -ENODEV;
    HIP_RET(hipGetDevice(&d));
    return d;
}

int spmv_device_info(int device, char *name, size_t len, int *compute_units,
                     size_t *hbm_bytes) {
    if (spmv_device_count() == 0)
        return -ENODEV;
    hipDeviceProp_t p;
    HIP_RET(hipGetDeviceProperties(&p, device));
    if (name && len)
        snprintf(name, len, "%s (%s)", p.name, p.gcnArchName);
    if (compute_units)
        *compute_units = p.multiProcessorCount;
    if (hbm_bytes)
        *hbm_bytes = p.totalGlobalMem;
    return 0;
}

/* "0000:c1:00.0": which physical card a rank drives (multi-GPU lines) */
int spmv_device_pci_bus_id(int device, char *buf, size_t len) {
    if (!buf || len < 16)
        return -EINVAL;
    if (spmv_device_count() == 0)
        return -ENODEV;
    HIP_RET(hipDeviceGetPCIBusId(buf, (int)len, device));
    return 0;
}

int spmv_dev_mem_info(size_t *free_bytes, size_t *total_bytes) {
    if (spmv_device_count() == 0)
        return -ENODEV;
    size_t f = 0, t = 0;
    HIP_RET(hipMemGetInfo(&f, &t));
    if (free_bytes)
        *free_bytes = f;
    if (total_bytes)
        *total_bytes = t;
    return 0;
}

int spmv_dev_malloc(void **dptr, size_t bytes) {
    if (!dptr)
        return -EINVAL;
    *dptr = NULL;
    if (spmv_device_count() == 0)
        return -ENODEV;
    HIP_RET(hipMalloc(dptr, bytes ? bytes : 16));
    return 0;
}

int spmv_dev_free(void *dptr) {
    if (!dptr)
        return 0;
    HIP_RET(hipFree(dptr));
    return 0;
}

int spmv_dev_memset(void *dptr, int byte, size_t bytes, void *stream) {
    HIP_RET(hipMemsetAsync(dptr, byte, bytes, (hipStream_t)stream));
    return 0;
}

int spmv_copy_h2d(void *dst, const void *src, size_t bytes) {
    if (bytes == 0)
        return 0;
    HIP_RET(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}

int spmv_copy_d2h(void *dst, const void *src, size_t bytes) {
    if (bytes == 0)
        return 0;
    HIP_RET(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int spmv_stream_sync(void *stream) {
    HIP_RET(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int spmv_device_sync(void) {
    HIP_RET(hipDeviceSynchronize());
    return 0;
}

/* the reference's GPU timer (cuda_timer.cu:15-21) as plain handles: an event
 * pair recorded on the launch stream around whatever the caller enqueues */
int spmv_event_create(void **ev) {
    if (!ev)
        return -EINVAL;
    hipEvent_t e = NULL;
    HIP_RET(hipEventCreate(&e));
    *ev = (void *)e;
    return 0;
}

int spmv_event_record(void *ev, void *stream) {
    if (!ev)
        return -EINVAL;
    HIP_RET(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return 0;
}

int spmv_event_elapsed_ms(void *start, void *stop, float *ms) {
    if (!start || !stop || !ms)
        return -EINVAL;
    HIP_RET(hipEventSynchronize((hipEvent_t)stop));
    HIP_RET(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return 0;
}

int spmv_event_destroy(void *ev) {
    if (ev)
        HIP_RET(hipEventDestroy((hipEvent_t)ev));
    return 0;
}

/* ---- streams and hipGraph capture of stream-ordered launches ---- */
int spmv_stream_create(void **stream) {
    if (!stream)
        return -EINVAL;
    hipStream_t s = NULL;
    HIP_RET(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return 0;
}

int spmv_stream_destroy(void *stream) {
    if (stream)
        HIP_RET(hipStreamDestroy((hipStream_t)stream));
    return 0;
}

int spmv_graph_begin_capture(void *stream) {
    if (!stream)
        return -EINVAL; /* the legacy default stream cannot be captured */
    HIP_RET(hipStreamBeginCapture((hipStream_t)stream,
                                  hipStreamCaptureModeThreadLocal));
    return 0;
}

int spmv_graph_end_capture(void *stream, void **graph_exec) {
    if (!stream || !graph_exec)
        return -EINVAL;
    hipGraph_t g = NULL;
    hipGraphExec_t e = NULL;
    HIP_RET(hipStreamEndCapture((hipStream_t)stream, &g));
    hipError_t err = hipGraphInstantiate(&e, g, NULL, NULL, 0);
    (void)hipGraphDestroy(g);
    if (err != hipSuccess)
        return hip_errno(err);
    *graph_exec = (void *)e;
    return 0;
}

int spmv_graph_launch(void *graph_exec, void *stream) {
    if (!graph_exec)
        return -EINVAL;
    HIP_RET(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return 0;
}

int spmv_graph_destroy(void *graph_exec) {
    if (graph_exec)
        HIP_RET(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return 0;
}

} /* extern "C" */

/* ------------------------------------------------------------------ */
/* device-side synthetic generation (include/spmv_synth.h)              */
/* ------------------------------------------------------------------ */

__global__ void k_fill_x(double *x, int64_t n, uint64_t seed, int64_t first) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        x[i] = synth_x(seed, first + i);
}

__global__ void k_synth_lens(synth_spec s, int *lens) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < s.M)
        lens[i] = synth_row_len(&s, s.row0 + i);
}

/* Rows of the power-law family and the hub row draw every column on its own
 * (synth_col_strat: stratified, ascending by construction), so a long one
 * can be filled by many lanes instead of one walking 10^5..10^8 entries (a
 * hub row of 131 072 entries was 79 ms of generation, one of 2.7e8 three
 * minutes) -- the same functions per entry, the same matrix. */
#define SYNTH_PARALLEL_ROW 4096
__host__ __device__ static inline bool synth_row_is_parallel(const synth_spec *s,
                                                             int64_t g, int len) {
    return len > SYNTH_PARALLEL_ROW &&
           (s->kind == SYNTH_POWERLAW ||
            (s->kind == SYNTH_HUB && g == synth_hub_row(s)));
}

/*
 * One lane per row, one workgroup per SYNTH_WG consecutive rows -- whose
 * entries are ONE contiguous range of JA / AS.  The range is generated in
 * LDS, every row at its final position relative to the range start (a padding
 * word per 32 -- SYNTH_AT -- keeps lanes that walk rows of 32 entries out of
 * each other's bank; the insertion sort of the random families runs there),
 * and then written out by the whole workgroup with consecutive lanes on
 * consecutive entries.  Round 4 let every lane write its own row straight to
 * HBM: 64 lanes x 4-byte stores to 64 different lines per instruction --
 * 130 GB of write traffic for a 3.84 GB matrix, 36 ms
 * (profiles/r04_wn_hll_tile_panels.md).  A range that does not fit the LDS
 * budget (`cap` entries: long rows among the 128) falls back to direct
 * stores; the same functions per entry either way, the same matrix.
 */
#define SYNTH_WG 128
#define SYNTH_SKEW 5
__global__ void __launch_bounds__(SYNTH_WG)
    k_synth_rows(synth_spec s, const int *__restrict__ irp, int *ja, double *as,
                 int cap) {
    extern __shared__ double synth_lds[];
    const int r0 = blockIdx.x * SYNTH_WG;
    const int r1 = min(r0 + SYNTH_WG, s.M);
    const int i = r0 + (int)threadIdx.x;
    const int beg0 = irp[r0];
    const int total = irp[r1] - beg0;
    int beg = 0, len = 0;
    bool mine = false;
    if (i < r1) {
        beg = irp[i];
        len = irp[i + 1] - beg;
        mine = !synth_row_is_parallel(&s, s.row0 + i, len); /* k_synth_long_rows */
    }
    /* A row left to k_synth_long_rows (longer than SYNTH_PARALLEL_ROW = 4096
     * entries, which can be BELOW cap: up to 12288) would leave its part of
     * the staged range unwritten, and the write-out loop would copy that
     * uninitialised LDS to ja / as -- harmless only as long as
     * k_synth_long_rows runs later on the same stream and overwrites it
     * (ADVICE r05).  No such ordering dependency: a workgroup that holds such
     * a row is not staged. */
    const int skipped = __syncthreads_or(i < r1 && !mine);
    if (total > cap || skipped) { /* workgroup-uniform: direct stores */
        if (mine)
            synth_fill_row(&s, s.row0 + i, len, ja + beg, as + beg);
        return;
    }
    double *vals = synth_lds;
    int *cols = (int *)(synth_lds + SYNTH_AT(cap, SYNTH_SKEW) + 1);
    if (mine)
        synth_fill_row_at(&s, s.row0 + i, len, cols, vals, beg - beg0,
                          SYNTH_SKEW);
    __syncthreads();
    /* every row of the range was generated above (none left to
     * k_synth_long_rows in a staged workgroup) */
    for (int p = (int)threadIdx.x; p < total; p += SYNTH_WG) {
        ja[beg0 + p] = cols[SYNTH_AT(p, SYNTH_SKEW)];
        as[beg0 + p] = vals[SYNTH_AT(p, SYNTH_SKEW)];
    }
}

/* grid (long rows, chunks): the workgroups of a row stride over its entries */
__global__ void k_synth_long_rows(synth_spec s, const int *irp,
                                  const int *rows, int *ja, double *as) {
    const int i = rows[blockIdx.x];
    const int64_t g = s.row0 + i;
    const int beg = irp[i], len = irp[i + 1] - beg;
    int64_t lo = 0, hi = s.N;
    if (s.kind != SYNTH_HUB)
        synth_window(&s, g, &lo, &hi);
    int *cols = ja + beg;
    double *vals = as + beg;
    for (int64_t j = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; j < len;
         j += (int64_t)gridDim.y * blockDim.x) {
        cols[j] = synth_col_strat(&s, g, (int)j, len, lo, hi);
        vals[j] = synth_val(&s, g, (int)j);
    }
}

/* CSR -> HLL on the device: per-block width, then slot fill */
__global__ void k_block_width(int M, const int *irp, int *width) {
    int row = blockIdx.x * blockDim.x + threadIdx.x;
    int len = row < M ? irp[row + 1] - irp[row] : 0;
#pragma unroll
    for (int d = 16; d > 0; d >>= 1)
        len = max(len, __shfl_down(len, d, 32));
    if ((threadIdx.x & 31) == 0 && row < M)
        width[row / 32] = len;
}

/* a hack block is as wide as its longest row: with one lane per row the 32
 * lanes of a hub block walk 10^5..10^8 slots each (93 ms of a 1M-row
 * conversion).  Blocks beyond HLL_FILL_WIDE columns are filled a lane per
 * SLOT instead (grid: wide blocks x chunks) -- same slots, same pad rule */
#define HLL_FILL_WIDE 2048
__global__ void k_hll_fill_wide(int M, int col_major, const int *wide_blocks,
                                const int *irp, const int *cja,
                                const double *cas, const int64_t *off, int *ja,
                                double *as, unsigned *padmask) {
    const int b = wide_blocks[blockIdx.x];
    const int rows = min(32, M - b * 32);
    const int64_t o = off[b];
    const int64_t n = off[b + 1] - o;
    const int w = (int)(n / rows);
    for (int64_t q = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; q < n;
         q += (int64_t)gridDim.y * blockDim.x) {
        const int i = col_major ? (int)(q % rows) : (int)(q / w);
        const int j = col_major ? (int)(q / rows) : (int)(q % w);
        const int beg = irp[b * 32 + i], len = irp[b * 32 + i + 1] - beg;
        const int64_t t = o + q;
        if (j < len) {
            ja[t] = cja[beg + j];
            as[t] = cas[beg + j];
        } else { /* pad -> the row's last valid column, or 0 (hip_hll.h) */
            ja[t] = len > 0 ? cja[beg + len - 1] : 0;
            as[t] = 0.0;
            atomicOr(padmask + (t >> 5), 1u << (t & 31));
        }
    }
}

/*
 * HLL_FILL_BLOCKS hack blocks (128 rows) per workgroup.  Their CSR entries
 * are one contiguous range of cja / cas: it is staged in LDS with whole-line
 * loads (skewed a word per 32 so that lanes reading 32 different rows'
 * j-th entries do not meet in one bank), then the workgroup walks the SLOTS
 * of its blocks, consecutive lanes on consecutive slots in either layout --
 * whole-line stores -- taking each slot's entry out of LDS.  Round 4's lane
 * per row read cja / cas with a stride of the row length: 65.9 GB read to
 * write 3.84 GB (profiles/r04_wn_hll_tile_panels.md).  A group whose entries
 * exceed the LDS budget reads them from global memory instead (same slot
 * walk); blocks wider than HLL_FILL_WIDE are k_hll_fill_wide's.  The pad
 * bits of 64 consecutive slots are OR-ed in with at most three atomics per
 * wavefront.
 */
#define HLL_FILL_BLOCKS 4
#define HLL_FILL_SKEW 5
__global__ void __launch_bounds__(256)
    k_hll_fill(int M, int nb, int col_major, const int *__restrict__ irp,
               const int *__restrict__ cja, const double *__restrict__ cas,
               const int64_t *__restrict__ off, int *ja, double *as,
               unsigned *padmask, int cap) {
    extern __shared__ double fill_lds[];
    __shared__ int s_irp[HLL_FILL_BLOCKS * 32 + 1];
    const int b0 = blockIdx.x * HLL_FILL_BLOCKS;
    const int b1 = min(b0 + HLL_FILL_BLOCKS, nb);
    const int r0 = b0 * 32, r1 = min(b1 * 32, M);
    for (int k = (int)threadIdx.x; k <= r1 - r0; k += 256)
        s_irp[k] = irp[r0 + k];
    __syncthreads();
    const int beg0 = s_irp[0];
    const int total = s_irp[r1 - r0] - beg0;
    const bool staged = total <= cap;
    double *vals = fill_lds;
    int *cols = (int *)(fill_lds + SYNTH_AT(cap, HLL_FILL_SKEW) + 1);
    if (staged) {
        for (int p = (int)threadIdx.x; p < total; p += 256) {
            cols[SYNTH_AT(p, HLL_FILL_SKEW)] = cja[beg0 + p];
            vals[SYNTH_AT(p, HLL_FILL_SKEW)] = cas[beg0 + p];
        }
        __syncthreads();
    }
    const int lane = (int)threadIdx.x & 63;
    for (int b = b0; b < b1; ++b) {
        const int rows = min(32, M - b * 32);
        const int64_t o = off[b];
        const int64_t n = off[b + 1] - o;
        const int w = (int)(n / rows);
        if (w > HLL_FILL_WIDE)
            continue; /* k_hll_fill_wide */
        /* whole wavefronts enter every round: the pad ballot needs all lanes */
        for (int64_t q0 = 0; q0 < n; q0 += 256) {
            const int64_t q = q0 + (int64_t)threadIdx.x;
            bool pad = false;
            if (q < n) {
                const int i = col_major ? (int)(q % rows) : (int)(q / w);
                const int j = col_major ? (int)(q / rows) : (int)(q % w);
                const int rb = s_irp[(b - b0) * 32 + i] - beg0;
                const int len = s_irp[(b - b0) * 32 + i + 1] - beg0 - rb;
                /* pad -> the row's last valid column, or 0 (hip_hll.h) */
                pad = j >= len;
                const int p = rb + (pad ? len - 1 : j);
                int c = 0;
                double v = 0.0;
                if (p >= rb) { /* len > 0 */
                    c = staged ? cols[SYNTH_AT(p, HLL_FILL_SKEW)] : cja[beg0 + p];
                    if (!pad)
                        v = staged ? vals[SYNTH_AT(p, HLL_FILL_SKEW)]
                                   : cas[beg0 + p];
                }
                ja[o + q] = c;
                as[o + q] = v;
            }
            const unsigned long long m = __ballot(pad);
            if (m) { /* slots t0 .. t0 + 63 of this wavefront: <= 3 words */
                const int64_t t0 = o + q0 + ((int64_t)threadIdx.x & ~63);
                const int sh = (int)(t0 & 31);
                unsigned word = 0;
                if (lane == 0)
                    word = (unsigned)(m << sh);
                else if (lane == 1)
                    word = (unsigned)(sh ? m >> (32 - sh) : m >> 32);
                else if (lane == 2)
                    word = sh ? (unsigned)(m >> (64 - sh)) : 0u;
                if (lane < 3 && word)
                    atomicOr(padmask + (t0 >> 5) + lane, word);
            }
        }
    }
}

/* scratch sweep used to push a small working set out of the Infinity Cache:
 * read-modify-write form (round 1 / 2; kept for A/B, variant bit 29) */
__global__ void k_flush(double *buf, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride)
        buf[i] = buf[i] * 1.0000001 + 1.0;
}

/* read-only form, the default since round 3: evicts just the same (reads
 * allocate in L2 and in the Infinity Cache) but leaves no DIRTY lines behind.
 * The read-modify-write flush left up to 256 MiB of dirty scratch in the
 * caches, and its write-back competed with the reads of the launch being
 * timed -- traffic that belongs to the flush, not to the SpMV: 1M x 16
 * banded, flushed, same run (gpurun r3c2): CSR stream 0.0433 ms after the RMW
 * flush / 0.0423 read-only / 0.0369 not flushed; thread-per-row HLL 0.0411 /
 * 0.0375 / 0.0337; sub-wave CSR 0.0546 / 0.0492 / 0.0479. */
__global__ void k_flush_ro(const double *buf, size_t n, double *sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (; i < n; i += stride)
        acc += buf[i];
    if (acc == 1.2345e300) /* never: the buffer holds zeros */
        *sink = acc;
}

/* ------------------------------------------------------------------ */
/* row-block table of the CSR stream kernel (host, O(M))                */
/* ------------------------------------------------------------------ */
static int finish_csr_handle(spmv_csr_dev *d, const int *host_irp) {
    int rc = 0;
    std::vector<int> tab, ent;
    std::vector<unsigned char> mode;
    std::vector<int> tmp;
    bool segs = false;
    if (!host_irp) {
        tmp.resize((size_t)d->M + 1);
        HIP_TRY(hipMemcpy(tmp.data(), d->irp, ((size_t)d->M + 1) * sizeof(int),
                          hipMemcpyDeviceToHost));
        host_irp = tmp.data();
    }
    stream_table_build(host_irp, d->M, STREAM_NNZ, STREAM_ROWS, STREAM_ROW_T,
                       STREAM_LONG_ROW, STREAM_SEG, tab, ent, mode,
                       &d->max_row_len, &segs);
    {
        /* constant row length?  (O(M) over the host copy of IRP) */
        /* (IRP[r] = r * len is what the kernel then computes: only for an
         * IRP that starts at 0) */
        int len = d->M > 0 && host_irp[0] == 0 ? host_irp[1] - host_irp[0] : 0;
        for (int r = 1; r < d->M && len > 0; ++r)
            if (host_irp[r + 1] - host_irp[r] != len)
                len = 0;
        d->uniform_len = len;
    }
    d->n_rowblk = (int)tab.size() - 1;
    {
        /* (first row, first entry) per range: the kernel learns both with one
         * load instead of a load of the row and a dependent load of IRP */
        std::vector<int> tab2(tab.size() * 2);
        for (size_t k = 0; k < tab.size(); ++k) {
            tab2[2 * k] = tab[k];
            tab2[2 * k + 1] = ent[k];
        }
        HIP_TRY(hipMalloc((void **)&d->rowblk, tab2.size() * sizeof(int)));
        HIP_TRY(hipMemcpy(d->rowblk, tab2.data(), tab2.size() * sizeof(int),
                          hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc((void **)&d->rowblk_mode, mode.size()));
    HIP_TRY(hipMemcpy(d->rowblk_mode, mode.data(), mode.size(),
                      hipMemcpyHostToDevice));
    {
        /* ranges holding (a segment of) one row of more than STREAM_NNZ
         * entries: what kernels 0-3 leave to k_csr_long_seg */
        std::vector<int> lrb;
        for (int k = 0; k < d->n_rowblk; ++k)
            if (mode[(size_t)k] == 2 || ent[(size_t)k + 1] - ent[(size_t)k] > STREAM_NNZ)
                lrb.push_back(k);
        d->n_long_rb = (int)lrb.size();
        if (!lrb.empty()) {
            HIP_TRY(hipMalloc((void **)&d->long_rb, lrb.size() * sizeof(int)));
            HIP_TRY(hipMemcpy(d->long_rb, lrb.data(), lrb.size() * sizeof(int),
                              hipMemcpyHostToDevice));
        }
    }
    if (segs) { /* partial sums + arrival counters of the long rows' ranges */
        const size_t n = (size_t)d->n_rowblk + 1;
        HIP_TRY(hipMalloc((void **)&d->seg_partial, n * sizeof(double)));
        HIP_TRY(hipMalloc((void **)&d->seg_count,
                          n * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(d->seg_partial, 0, n * sizeof(double)));
        HIP_TRY(hipMemset(d->seg_count, 0, n * sizeof(unsigned long long)));
    }
fail:
    return rc;
}

extern "C" {

int spmv_dev_fill_synth(double *d_x, int64_t n, uint64_t seed, int64_t first,
                        void *stream) {
    if (n <= 0)
        return 0;
    hipLaunchKernelGGL(k_fill_x, dim3((unsigned)((n + 255) / 256)), dim3(256),
                       0, (hipStream_t)stream, d_x, n, seed, first);
    return hip_errno(hipGetLastError());
}

/* ------------------------------------------------------------------ */
/* CSR handle                                                           */
/* ------------------------------------------------------------------ */

static void csr_teardown(spmv_csr_dev *d) {
    (void)hipFree(d->irp);
    (void)hipFree(d->ja);
    (void)hipFree(d->as);
    (void)hipFree(d->rowblk);
    (void)hipFree(d->rowblk_mode);
    (void)hipFree(d->seg_partial);
    (void)hipFree(d->seg_count);
    (void)hipFree(d->long_rb);
    panels_free(d->panels);
    free(d->tune_log);
    free(d);
}

void spmv_csr_release(spmv_csr_dev *d) {
    if (d && live_take(d)) /* else: ignored, counted (spmv_ignored_releases) */
        csr_teardown(d);
}

void spmv_csr_release_checked(spmv_csr_dev *d, uint64_t generation) {
    if (d && generation && live_take(d, generation))
        csr_teardown(d);
}

static int csr_alloc_dev(int M, int N, int64_t NZ, spmv_csr_dev **out) {
    int rc = 0;
    spmv_csr_dev *d = (spmv_csr_dev *)calloc(1, sizeof *d);
    if (!d)
        return -ENOMEM;
    live_add(d);
    d->M = M;
    d->N = N;
    d->NZ = NZ;
    /* workgroup orders before tuning (csr_kernels.hip): grouped runs for
     * matrices of >= 2M rows (banded 10M x 32: stream kernel 0.652 ms
     * grouped / 0.683 hardware, sub-wave 0.843 / 0.905 / 0.885 contiguous),
     * below: hardware order for the stream kernel (1M x 16: 0.0462 vs
     * 0.0471), contiguous ranges for the sub-wave kernel (0.0536 vs 0.0548) */
    d->order = M >= 2000000 ? 2 : 1;
    d->stream_grouped = M >= 2000000;
    HIP_TRY(hipGetDevice(&d->device));
    HIP_TRY(hipMalloc((void **)&d->irp, ((size_t)M + 1) * sizeof(int)));
    /* + slack: the stream kernel's 16-byte loads of the last range may read
     * up to 2048 + 3 entries past NZ (never used) */
    HIP_TRY(hipMalloc((void **)&d->ja, ((size_t)NZ + 2056) * sizeof(int)));
    HIP_TRY(hipMalloc((void **)&d->as, ((size_t)NZ + 2056) * sizeof(double)));
    *out = d;
    return 0;
fail:
    spmv_csr_release(d);
    return rc;
}

int spmv_csr_upload(const sparse_csr *A, spmv_csr_dev **out) {
    if (!A || !out || A->M < 0 || A->NZ < 0)
        return -EINVAL;
    *out = NULL;
    if (spmv_device_count() == 0)
        return -ENODEV;
    spmv_csr_dev *d = NULL;
    int rc = csr_alloc_dev(A->M, A->N, A->NZ, &d);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpy(d->irp, A->IRP, ((size_t)A->M + 1) * sizeof(int),
                      hipMemcpyHostToDevice));
    if (A->NZ > 0) {
        HIP_TRY(hipMemcpy(d->ja, A->JA, (size_t)A->NZ * sizeof(int),
                          hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d->as, A->AS, (size_t)A->NZ * sizeof(double),
                          hipMemcpyHostToDevice));
    }
    rc = finish_csr_handle(d, A->IRP);
    if (rc)
        goto fail;
    *out = d;
    return 0;
fail:
    spmv_csr_release(d);
    return rc;
}

int spmv_csr_generate(int kind, int M, int N, int K, int64_t W, int64_t row0,
                      uint64_t seed, spmv_csr_dev **out) {
    if (!out || M < 0 || N <= 0 || K <= 0 || kind < SYNTH_BANDED ||
        kind > SYNTH_KIND_LAST || (kind == SYNTH_BANDED && N < K))
        return -EINVAL;
    *out = NULL;
    if (spmv_device_count() == 0)
        return -ENODEV;
    synth_spec s = {kind, M, N, K, W, row0, seed};
    int rc = 0;
    spmv_csr_dev *d = NULL;
    /* row lengths on the device, prefix sum on the host (O(M) ints) */
    std::vector<int> irp((size_t)M + 1, 0);
    if (M > 0) {
        int *d_len = NULL;
        HIP_RET(hipMalloc((void **)&d_len, (size_t)M * sizeof(int)));
        hipLaunchKernelGGL(k_synth_lens, dim3((M + 255) / 256), dim3(256), 0,
                           0, s, d_len);
        hipError_t e = hipMemcpy(irp.data() + 1, d_len, (size_t)M * sizeof(int),
                                 hipMemcpyDeviceToHost);
        (void)hipFree(d_len);
        if (e != hipSuccess)
            return hip_errno(e);
    }
    int64_t nz = 0;
    for (int i = 0; i < M; ++i) {
        nz += irp[(size_t)i + 1];
        if (nz > INT32_MAX)
            return -EOVERFLOW;
        irp[(size_t)i + 1] = (int)nz;
    }
    rc = csr_alloc_dev(M, N, nz, &d);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpy(d->irp, irp.data(), ((size_t)M + 1) * sizeof(int),
                      hipMemcpyHostToDevice));
    if (M > 0) {
        /* LDS budget of a 128-row range: the nominal row length with a
         * quarter of slack (the ragged family's longest row) + 8, at most
         * 12288 entries (147 KiB); heavier ranges store directly */
        const int per_row = K + K / 4 + 8;
        const int cap = (int)std::min<long long>(12288, (long long)SYNTH_WG * per_row);
        const size_t lds = ((size_t)SYNTH_AT(cap, SYNTH_SKEW) + 1) * 12 + 16;
        if (lds > 64 * 1024) /* per device, per call: a one-off setup path */
            HIP_TRY(hipFuncSetAttribute(
                (const void *)k_synth_rows,
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_synth_rows, dim3((M + SYNTH_WG - 1) / SYNTH_WG),
                           dim3(SYNTH_WG), lds, 0, s, d->irp, d->ja, d->as, cap);
    }
    HIP_TRY(hipGetLastError());
    {
        std::vector<int> long_rows;
        int longest = 0;
        for (int i = 0; i < M; ++i) {
            const int len = irp[(size_t)i + 1] - irp[i];
            if (synth_row_is_parallel(&s, row0 + i, len)) {
                long_rows.push_back(i);
                longest = std::max(longest, len);
            }
        }
        if (!long_rows.empty()) {
            int *d_rows = NULL;
            HIP_TRY(hipMalloc((void **)&d_rows, long_rows.size() * sizeof(int)));
            hipError_t e = hipMemcpy(d_rows, long_rows.data(),
                                     long_rows.size() * sizeof(int),
                                     hipMemcpyHostToDevice);
            if (e == hipSuccess) {
                const int chunks = std::min(1024, (longest + 4095) / 4096);
                hipLaunchKernelGGL(k_synth_long_rows,
                                   dim3((unsigned)long_rows.size(), chunks),
                                   dim3(256), 0, 0, s, d->irp, d_rows, d->ja,
                                   d->as);
                e = hipGetLastError();
                if (e == hipSuccess)
                    e = hipDeviceSynchronize();
            }
            (void)hipFree(d_rows);
            if (e != hipSuccess) {
                rc = hip_errno(e);
                goto fail;
            }
        }
    }
    HIP_TRY(hipDeviceSynchronize());
    rc = finish_csr_handle(d, irp.data());
    if (rc)
        goto fail;
    *out = d;
    return 0;
fail:
    spmv_csr_release(d);
    return rc;
}

int spmv_csr_shape(const spmv_csr_dev *A, int *M, int *N, int64_t *NZ) {
    HANDLE_OK(A);
    if (M)
        *M = A->M;
    if (N)
        *N = A->N;
    if (NZ)
        *NZ = A->NZ;
    return 0;
}

int64_t spmv_csr_algorithmic_bytes(const spmv_csr_dev *A) {
    HANDLE_OK(A);
    return 12 * A->NZ + 4 * ((int64_t)A->M + 1) + 8 * (int64_t)A->M +
           8 * (int64_t)A->N;
}

int spmv_csr_download(const spmv_csr_dev *A, sparse_csr **out) {
    HANDLE_OK(A);
    if (!out)
        return -EINVAL;
    if (!A->ja && A->NZ > 0)
        return -ENODATA; /* spmv_csr_release_source() */
    if (A->NZ > INT32_MAX)
        return -EOVERFLOW;
    sparse_csr *h = csr_alloc("device", A->M, A->N, (int)A->NZ);
    if (IS_ERR(h))
        return PTR_ERR(h);
    int rc = 0;
    HIP_TRY(hipMemcpy(h->IRP, A->irp, ((size_t)A->M + 1) * sizeof(int),
                      hipMemcpyDeviceToHost));
    if (A->NZ > 0) {
        HIP_TRY(hipMemcpy(h->JA, A->ja, (size_t)A->NZ * sizeof(int),
                          hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(h->AS, A->as, (size_t)A->NZ * sizeof(double),
                          hipMemcpyDeviceToHost));
    }
    *out = h;
    return 0;
fail:
    csr_free(h);
    return rc;
}

int spmv_csr_build_panels(spmv_csr_dev *A, int panel_cols) {
    HANDLE_OK(A);
    if (!A->ja && A->NZ > 0)
        return -ENODATA;
    panels_free(A->panels);
    A->panels = NULL;
    return panels_from_csr(A, panel_cols, -1, 0, &A->panels);
}

int spmv_hll_build_panels(spmv_hll_dev *H, int panel_cols) {
    HANDLE_OK(H);
    if (!H->ja && H->slots > 0)
        return -ENODATA;
    panels_free(H->panels);
    H->panels = NULL;
    return panels_from_hll(H, panel_cols, -1, 0, &H->panels);
}

int spmv_csr_build_panels_opts(spmv_csr_dev *A, const spmv_panel_opts *opts) {
    HANDLE_OK(A);
    if (!A->ja && A->NZ > 0)
        return -ENODATA;
    panels_free(A->panels);
    A->panels = NULL;
    return panels_from_csr_opts(A, opts, &A->panels);
}

int spmv_hll_build_panels_opts(spmv_hll_dev *H, const spmv_panel_opts *opts) {
    HANDLE_OK(H);
    if (!H->ja && H->slots > 0)
        return -ENODATA;
    panels_free(H->panels);
    H->panels = NULL;
    return panels_from_hll_opts(H, opts, &H->panels);
}

/* same schedule and tile height as `model`'s blocked copy (shards of one
 * matrix: tune one, build the others alike) */
int spmv_csr_build_panels_like(spmv_csr_dev *A, const spmv_csr_dev *model) {
    HANDLE_OK(A);
    HANDLE_OK(model);
    if (!model->panels)
        return -EINVAL;
    if (!A->ja && A->NZ > 0)
        return -ENODATA;
    panels_free(A->panels);
    A->panels = NULL;
    spmv_panel_opts o;
    panels_get_opts(model->panels, &o);
    int rc = panels_from_csr_opts(A, &o, &A->panels);
    if (!rc)
        panels_set_waves(A->panels, panels_waves(model->panels));
    return rc;
}

int spmv_hll_build_panels_like(spmv_hll_dev *H, const spmv_hll_dev *model) {
    HANDLE_OK(H);
    HANDLE_OK(model);
    if (!model->panels)
        return -EINVAL;
    if (!H->ja && H->slots > 0)
        return -ENODATA;
    panels_free(H->panels);
    H->panels = NULL;
    spmv_panel_opts o;
    panels_get_opts(model->panels, &o);
    int rc = panels_from_hll_opts(H, &o, &H->panels);
    if (!rc)
        panels_set_waves(H->panels, panels_waves(model->panels));
    return rc;
}

static int panels_info(const spmv_panels *P, int *steps, int *tiles,
                       int *panels, int64_t *entries) {
    if (!P)
        return -ENOENT; /* not built */
    if (steps)
        *steps = panels_steps(P);
    if (tiles)
        *tiles = panels_tiles(P);
    if (panels)
        *panels = panels_count(P);
    if (entries)
        *entries = panels_nnz(P);
    return 0;
}

int spmv_csr_panels_info(const spmv_csr_dev *A, int *steps, int *tiles,
                         int *panels, int64_t *entries) {
    HANDLE_OK(A);
    return panels_info(A->panels, steps, tiles, panels, entries);
}

/*
 * Keep only the blocked copy: frees JA/AS (12 B per entry), so a handle that
 * runs the blocked path costs the same HBM as the format it came from.
 * Afterwards only the PANELS kernel id can be launched; the direct kernels,
 * download, conversion and further build_panels calls return -ENODATA.
 */
int spmv_csr_release_source(spmv_csr_dev *A) {
    HANDLE_OK(A);
    if (!A->panels)
        return -ENOENT; /* nothing else could run the matrix */
    (void)hipFree(A->ja);
    (void)hipFree(A->as);
    A->ja = NULL;
    A->as = NULL;
    return 0;
}

int spmv_hll_release_source(spmv_hll_dev *H) {
    HANDLE_OK(H);
    if (!H->panels)
        return -ENOENT;
    (void)hipFree(H->ja);
    (void)hipFree(H->as);
    (void)hipFree(H->padmask);
    H->ja = NULL;
    H->as = NULL;
    H->padmask = NULL;
    return 0;
}

static int panels_schedule_of(const spmv_panels *P) {
    if (!P)
        return -ENOENT;
    return panels_is_sweep(P) ? 1 : panels_is_chain(P) ? 2 : 0;
}

int spmv_csr_panels_schedule(const spmv_csr_dev *A) {
    HANDLE_OK(A);
    return panels_schedule_of(A->panels);
}

int spmv_hll_panels_schedule(const spmv_hll_dev *H) {
    HANDLE_OK(H);
    return panels_schedule_of(H->panels);
}

int spmv_csr_panels_describe(const spmv_csr_dev *A, char *buf, size_t len) {
    HANDLE_OK(A);
    return A->panels ? panels_describe(A->panels, buf, len) : -ENOENT;
}

int spmv_hll_panels_describe(const spmv_hll_dev *H, char *buf, size_t len) {
    HANDLE_OK(H);
    return H->panels ? panels_describe(H->panels, buf, len) : -ENOENT;
}

int spmv_csr_panels_tile_rows(const spmv_csr_dev *A) {
    HANDLE_OK(A);
    return A->panels ? panels_tile_rows(A->panels) : -ENOENT;
}

int spmv_hll_panels_tile_rows(const spmv_hll_dev *H) {
    HANDLE_OK(H);
    return H->panels ? panels_tile_rows(H->panels) : -ENOENT;
}

/* The layout of the blocked copy as build options + the launch's waves hint:
 * build_panels_opts(o) followed by panels_set_waves(waves) on another handle
 * of the same matrix reproduces exactly what the selector settled on (the
 * profiling passes of a workload pin the layout of the un-profiled run this
 * way: under the counters' serialised launches the selector may pick another
 * candidate).  The caller sets o->struct_size = sizeof *o first. */
static int panels_layout_of(const spmv_panels *P, spmv_panel_opts *o,
                            int *waves) {
    if (!o || o->struct_size != (int)sizeof *o)
        return -EINVAL;
    if (!P)
        return -ENOENT;
    panels_get_opts(P, o);
    if (waves)
        *waves = panels_waves(P);
    return 0;
}

int spmv_csr_panels_layout(const spmv_csr_dev *A, spmv_panel_opts *o, int *waves) {
    HANDLE_OK(A);
    return panels_layout_of(A->panels, o, waves);
}

int spmv_hll_panels_layout(const spmv_hll_dev *H, spmv_panel_opts *o, int *waves) {
    HANDLE_OK(H);
    return panels_layout_of(H->panels, o, waves);
}

/* waves per workgroup of the blocked launch (0: the kernel's default) */
int spmv_csr_panels_set_waves(spmv_csr_dev *A, int waves) {
    HANDLE_OK(A);
    if (!A->panels)
        return -ENOENT;
    if (waves < 0 || waves > 16)
        return -EINVAL;
    panels_set_waves(A->panels, waves);
    return 0;
}

int spmv_hll_panels_set_waves(spmv_hll_dev *H, int waves) {
    HANDLE_OK(H);
    if (!H->panels)
        return -ENOENT;
    if (waves < 0 || waves > 16)
        return -EINVAL;
    panels_set_waves(H->panels, waves);
    return 0;
}

/* explicit schedule (0 steps, 1 sweep, 2 chain) and tile height (0: default;
 * ignored by sweep): ranks of a multi-GPU job build what rank 0 tuned */
int spmv_csr_build_panels_as(spmv_csr_dev *A, int panel_cols, int sched,
                             int tile_rows) {
    HANDLE_OK(A);
    if (sched < 0 || sched > 2)
        return -EINVAL;
    if (!A->ja && A->NZ > 0)
        return -ENODATA;
    panels_free(A->panels);
    A->panels = NULL;
    return panels_from_csr(A, panel_cols, sched, tile_rows, &A->panels);
}

int spmv_hll_build_panels_as(spmv_hll_dev *H, int panel_cols, int sched,
                             int tile_rows) {
    HANDLE_OK(H);
    if (sched < 0 || sched > 2)
        return -EINVAL;
    if (!H->ja && H->slots > 0)
        return -ENODATA;
    panels_free(H->panels);
    H->panels = NULL;
    return panels_from_hll(H, panel_cols, sched, tile_rows, &H->panels);
}

int spmv_hll_panels_info(const spmv_hll_dev *H, int *steps, int *tiles,
                         int *panels, int64_t *entries) {
    HANDLE_OK(H);
    return panels_info(H->panels, steps, tiles, panels, entries);
}

int spmv_csr_launch_rows(const spmv_csr_dev *A, int kernel,
                         const spmv_launch_opts *opts, const double *d_x,
                         double *d_y, int row_begin, int row_end,
                         void *stream) {
    HANDLE_OK(A);
    if (kernel == SPMV_CSR_KERNEL_PANELS) {
        if (!A->panels || row_begin != 0 || row_end != A->M)
            return -EINVAL; /* build panels first; whole matrix only */
        return panels_launch(A->panels, A->M,
                             opts ? opts->waves_per_block : 0,
                             opts ? opts->variant : 0, d_x, d_y,
                             (hipStream_t)stream);
    }
    if (!A->ja && A->NZ > 0)
        return -ENODATA; /* spmv_csr_release_source(): blocked path only */
    return csr_launch_kernel(A, kernel,
                             pick_waves(opts, default_waves(g_csr_waves, A->M)),
                             opts ? opts->group : 0, opts ? opts->variant : 0,
                             d_x, d_y, row_begin,
                             row_end, (hipStream_t)stream);
}

int spmv_csr_launch(const spmv_csr_dev *A, int kernel,
                    const spmv_launch_opts *opts, const double *d_x,
                    double *d_y, void *stream) {
    HANDLE_OK(A);
    return spmv_csr_launch_rows(A, kernel, opts, d_x, d_y, 0, A->M, stream);
}

/* ------------------------------------------------------------------ */
/* HLL handle                                                           */
/* ------------------------------------------------------------------ */

static void hll_teardown(spmv_hll_dev *d) {
    (void)hipFree(d->ja);
    (void)hipFree(d->as);
    (void)hipFree(d->off);
    (void)hipFree(d->padmask);
    (void)hipFree(d->wide_seg);
    (void)hipFree(d->wide_part);
    (void)hipFree(d->wide_cnt);
    panels_free(d->panels);
    free(d->tune_log);
    free(d);
}

void spmv_hll_release(spmv_hll_dev *d) {
    if (d && live_take(d))
        hll_teardown(d);
}

void spmv_hll_release_checked(spmv_hll_dev *d, uint64_t generation) {
    if (d && generation && live_take(d, generation))
        hll_teardown(d);
}

static int hll_alloc_dev(int M, int N, int64_t NZ, int nb, int col_major,
                         const int64_t *host_off, spmv_hll_dev **out) {
    int rc = 0;
    spmv_hll_dev *d = (spmv_hll_dev *)calloc(1, sizeof *d);
    if (!d)
        return -ENOMEM;
    live_add(d);
    d->M = M;
    d->N = N;
    d->NZ = NZ;
    d->nb = nb;
    d->col_major = col_major ? 1 : 0;
    d->slots = host_off[nb];
    /* before tuning: grouped order for matrices of >= 2M rows (banded
     * 10M x 32: 0.589 ms grouped / 0.622 hardware / 0.651 contiguous),
     * hardware order below (1M x 16: 0.0399 / 0.0426 / 0.0421) */
    d->order = nb >= 65536 ? 2 : 0;
    /* per XCD a contiguous run of hack blocks holding about 1/8 of the slots
     * (each block also counts 32 slots, so empty blocks spread evenly) */
    d->xcd_blk.first[0] = 0;
    for (int k = 1; k < NUM_XCD; ++k) {
        const double want =
            ((double)host_off[nb] + 32.0 * nb) * k / NUM_XCD;
        int lo = d->xcd_blk.first[k - 1], hi = nb;
        while (lo < hi) {
            const int mid = lo + (hi - lo) / 2;
            if ((double)host_off[mid] + 32.0 * mid < want)
                lo = mid + 1;
            else
                hi = mid;
        }
        d->xcd_blk.first[k] = (lo + 1) & ~1; /* block pairs stay together */
        if (d->xcd_blk.first[k] > nb)
            d->xcd_blk.first[k] = nb;
    }
    d->xcd_blk.first[NUM_XCD] = nb;
    HIP_TRY(hipGetDevice(&d->device));
    /* +64 slots of slack: vector loads of the last chunk stay in bounds */
    HIP_TRY(hipMalloc((void **)&d->ja, ((size_t)d->slots + 64) * sizeof(int)));
    HIP_TRY(hipMalloc((void **)&d->as, ((size_t)d->slots + 64) * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&d->off, ((size_t)nb + 1) * sizeof(int64_t)));
    {
        const size_t words = ((size_t)d->slots + 31) / 32 + 1;
        HIP_TRY(hipMalloc((void **)&d->padmask, words * sizeof(unsigned)));
        HIP_TRY(hipMemset(d->padmask, 0, words * sizeof(unsigned)));
    }
    HIP_TRY(hipMemcpy(d->off, host_off, ((size_t)nb + 1) * sizeof(int64_t),
                      hipMemcpyHostToDevice));
    {
        /* segments of the wide blocks (hip_common.h, HLL_WIDE / HLL_WSEG) */
        std::vector<int4> seg;
        for (int b = 0; b < nb; ++b) {
            const int rows = std::min(HACK_SIZE, M - b * HACK_SIZE);
            const int64_t w = rows > 0 ? (host_off[b + 1] - host_off[b]) / rows : 0;
            if (w <= HLL_WIDE)
                continue;
            /* segments of HLL_WSEG columns, at most HLL_WSEG_MAX of them:
             * wider ones then (multiples of 8 columns) */
            int nseg = (int)std::min<int64_t>((w + HLL_WSEG - 1) / HLL_WSEG,
                                              HLL_WSEG_MAX);
            const int segw = (int)(((w + nseg - 1) / nseg + 7) / 8 * 8);
            nseg = (int)((w + segw - 1) / segw);
            for (int k = 0; k < nseg; ++k)
                seg.push_back(make_int4(b, segw, k, nseg));
        }
        d->n_wide_seg = (int)seg.size();
        if (!seg.empty()) {
            HIP_TRY(hipMalloc((void **)&d->wide_seg, seg.size() * sizeof(int4)));
            HIP_TRY(hipMemcpy(d->wide_seg, seg.data(), seg.size() * sizeof(int4),
                              hipMemcpyHostToDevice));
            HIP_TRY(hipMalloc((void **)&d->wide_part,
                              seg.size() * HACK_SIZE * sizeof(double)));
            HIP_TRY(hipMalloc((void **)&d->wide_cnt,
                              seg.size() * sizeof(unsigned long long)));
            HIP_TRY(hipMemset(d->wide_part, 0,
                              seg.size() * HACK_SIZE * sizeof(double)));
            HIP_TRY(hipMemset(d->wide_cnt, 0,
                              seg.size() * sizeof(unsigned long long)));
        }
    }
    *out = d;
    return 0;
fail:
    spmv_hll_release(d);
    return rc;
}

int spmv_hll_upload(const sparse_hll *H, int is_col_major,
                    spmv_hll_dev **out) {
    if (!H || !out || H->hack_size != HACK_SIZE)
        return -EINVAL;
    *out = NULL;
    if (spmv_device_count() == 0)
        return -ENODEV;
    const int nb = H->num_blocks;
    std::vector<int64_t> off((size_t)nb + 1, 0);
    int maxw = 0;
    for (int b = 0; b < nb; ++b) {
        off[(size_t)b + 1] =
            off[b] + (int64_t)H->blocks[b].M * H->blocks[b].max_NZ;
        maxw = std::max(maxw, H->blocks[b].max_NZ);
    }
    spmv_hll_dev *d = NULL;
    int rc = hll_alloc_dev(H->M, H->N, H->NZ, nb, is_col_major, off.data(), &d);
    if (rc)
        return rc;
    d->max_width = maxw;
    if (d->slots > 0) {
        if (hll_is_contiguous(H)) { /* slab-backed: two copies */
            HIP_TRY(hipMemcpy(d->ja, H->blocks[0].JA,
                              (size_t)d->slots * sizeof(int),
                              hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(d->as, H->blocks[0].AS,
                              (size_t)d->slots * sizeof(double),
                              hipMemcpyHostToDevice));
        } else {
            /* blocks allocated one by one (the reference's csr_to_hll): pack
             * them into two host slabs in parallel, then two copies -- a copy
             * pair per block is 625 000 hipMemcpy calls at config 3.  Without
             * the host memory for the slabs: block by block after all. */
            int *pj = (int *)malloc((size_t)d->slots * sizeof(int));
            double *pa = (double *)malloc((size_t)d->slots * sizeof(double));
            if (pj && pa) {
                hll_pack_slabs(H, off.data(), pj, pa);
                hipError_t e = hipMemcpy(d->ja, pj, (size_t)d->slots * sizeof(int),
                                         hipMemcpyHostToDevice);
                if (e == hipSuccess)
                    e = hipMemcpy(d->as, pa, (size_t)d->slots * sizeof(double),
                                  hipMemcpyHostToDevice);
                free(pj);
                free(pa);
                HIP_TRY(e);
            } else {
                free(pj);
                free(pa);
                for (int b = 0; b < nb; ++b) {
                    size_t n = (size_t)(off[(size_t)b + 1] - off[b]);
                    if (!n)
                        continue;
                    HIP_TRY(hipMemcpy(d->ja + off[b], H->blocks[b].JA,
                                      n * sizeof(int), hipMemcpyHostToDevice));
                    HIP_TRY(hipMemcpy(d->as + off[b], H->blocks[b].AS,
                                      n * sizeof(double),
                                      hipMemcpyHostToDevice));
                }
            }
        }
        rc = hll_fix_pads_dev(d, 0);
        if (rc)
            goto fail;
        HIP_TRY(hipDeviceSynchronize());
    }
    *out = d;
    return 0;
fail:
    spmv_hll_release(d);
    return rc;
}

int spmv_hll_from_csr(const spmv_csr_dev *A, int is_col_major,
                      spmv_hll_dev **out) {
    HANDLE_OK(A);
    if (!out)
        return -EINVAL;
    if (!A->ja && A->NZ > 0)
        return -ENODATA;
    *out = NULL;
    const int M = A->M, nb = (M + 31) / 32;
    int rc = 0;
    int *d_w = NULL;
    spmv_hll_dev *d = NULL;
    std::vector<int> w((size_t)nb, 0);
    std::vector<int64_t> off((size_t)nb + 1, 0);
    int maxw = 0;
    if (nb > 0) {
        HIP_TRY(hipMalloc((void **)&d_w, (size_t)nb * sizeof(int)));
        hipLaunchKernelGGL(k_block_width, dim3((M + 255) / 256), dim3(256), 0,
                           0, M, A->irp, d_w);
        HIP_TRY(hipMemcpy(w.data(), d_w, (size_t)nb * sizeof(int),
                          hipMemcpyDeviceToHost));
    }
    for (int b = 0; b < nb; ++b) {
        int rows = std::min(32, M - b * 32);
        off[(size_t)b + 1] = off[b] + (int64_t)rows * w[b];
        maxw = std::max(maxw, w[b]);
    }
    rc = hll_alloc_dev(M, A->N, A->NZ, nb, is_col_major, off.data(), &d);
    if (rc)
        goto fail;
    d->max_width = maxw;
    if (M > 0) {
        {
            /* LDS budget of a group of 128 rows: mean row length + 25 % + 8
             * per row, 2048..12288 entries; heavier groups read global */
            const long long mean = M > 0 ? (A->NZ + M - 1) / M : 0;
            const int cap = (int)std::max<long long>(
                2048, std::min<long long>(
                          12288, HLL_FILL_BLOCKS * 32 * (mean + mean / 4 + 8)));
            const size_t lds = ((size_t)SYNTH_AT(cap, HLL_FILL_SKEW) + 1) * 12 + 16;
            if (lds > 64 * 1024)
                HIP_TRY(hipFuncSetAttribute(
                    (const void *)k_hll_fill,
                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_hll_fill,
                               dim3((nb + HLL_FILL_BLOCKS - 1) / HLL_FILL_BLOCKS),
                               dim3(256), lds, 0, M, nb, d->col_major, A->irp,
                               A->ja, A->as, d->off, d->ja, d->as, d->padmask,
                               cap);
        }
        HIP_TRY(hipGetLastError());
        std::vector<int> wide;
        for (int b = 0; b < nb; ++b)
            if (w[b] > HLL_FILL_WIDE)
                wide.push_back(b);
        if (!wide.empty()) {
            /* d_w (the widths) has done its job: reuse it for the list */
            HIP_TRY(hipMemcpy(d_w, wide.data(), wide.size() * sizeof(int),
                              hipMemcpyHostToDevice));
            const int64_t most = (int64_t)32 * maxw;
            const int chunks = (int)std::min<int64_t>(1024, (most + 8191) / 8192);
            hipLaunchKernelGGL(k_hll_fill_wide,
                               dim3((unsigned)wide.size(), chunks), dim3(256), 0,
                               0, M, d->col_major, d_w, A->irp, A->ja, A->as,
                               d->off, d->ja, d->as, d->padmask);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipDeviceSynchronize());
    }
    (void)hipFree(d_w);
    *out = d;
    return 0;
fail:
    (void)hipFree(d_w);
    spmv_hll_release(d);
    return rc;
}

int spmv_hll_shape(const spmv_hll_dev *H, int *M, int *N, int64_t *NZ,
                   int *num_blocks, int64_t *slots, int *is_col_major) {
    HANDLE_OK(H);
    if (M)
        *M = H->M;
    if (N)
        *N = H->N;
    if (NZ)
        *NZ = H->NZ;
    if (num_blocks)
        *num_blocks = H->nb;
    if (slots)
        *slots = H->slots;
    if (is_col_major)
        *is_col_major = H->col_major;
    return 0;
}

int64_t spmv_hll_algorithmic_bytes(const spmv_hll_dev *H) {
    HANDLE_OK(H);
    return 12 * H->slots + 12 * (int64_t)H->nb + 8 * (int64_t)H->M +
           8 * (int64_t)H->N;
}

/* what ONE launch of `kernel` has to move at least: the direct kernels read
 * every stored slot, padding included (12 S); the blocked copy (kernel 4)
 * holds the true entries only, so it is priced on NZ -- else a padded matrix
 * (S = 10 NZ on a power-law one) would show the blocked kernel above 100 % */
int64_t spmv_hll_kernel_bytes(const spmv_hll_dev *H, int kernel) {
    HANDLE_OK(H);
    if (kernel != SPMV_HLL_KERNEL_PANELS)
        return spmv_hll_algorithmic_bytes(H);
    return 12 * H->NZ + 12 * (int64_t)H->nb + 8 * (int64_t)H->M +
           8 * (int64_t)H->N;
}

int spmv_hll_launch_blocks(const spmv_hll_dev *H, int kernel,
                           const spmv_launch_opts *opts, const double *d_x,
                           double *d_y, int blk_begin, int blk_end,
                           void *stream) {
    HANDLE_OK(H);
    int waves = pick_waves(opts, default_waves(g_hll_waves, H->M));
    if (kernel == SPMV_HLL_KERNEL_PANELS) {
        if (!H->panels || blk_begin != 0 || blk_end != H->nb)
            return -EINVAL; /* build panels first; whole matrix only */
        return panels_launch(H->panels, H->M,
                             opts ? opts->waves_per_block : 0,
                             opts ? opts->variant : 0, d_x, d_y,
                             (hipStream_t)stream);
    }
    if (!H->ja && H->slots > 0)
        return -ENODATA; /* spmv_hll_release_source(): blocked path only */
    if (kernel == 1 && waves > 8)
        waves = 8; /* 6 KiB of LDS per wavefront, stay under 64 KiB */
    return hll_launch_kernel(H, kernel, waves, opts ? opts->variant : 0, d_x,
                             d_y, blk_begin, blk_end, (hipStream_t)stream);
}

int spmv_hll_launch(const spmv_hll_dev *H, int kernel,
                    const spmv_launch_opts *opts, const double *d_x,
                    double *d_y, void *stream) {
    HANDLE_OK(H);
    return spmv_hll_launch_blocks(H, kernel, opts, d_x, d_y, 0, H->nb, stream);
}

} /* extern "C" */

/* ------------------------------------------------------------------ */
/* event-timed loops                                                    */
/* ------------------------------------------------------------------ */

#define SPMV_VARIANT_FLUSH_RMW (1 << 29) /* A/B: the read-modify-write flush */

/* scratch buffer of the cache flush; a selector run owns one for all its
 * timed loops */
struct flush_scratch {
    double *buf = NULL;
    size_t bytes = 0;
    int reserve(size_t want) {
        if (want <= bytes)
            return 0;
        (void)hipFree(buf);
        buf = NULL;
        bytes = 0;
        HIP_RET(hipMalloc((void **)&buf, want));
        HIP_RET(hipMemset(buf, 0, want));
        bytes = want;
        return 0;
    }
    ~flush_scratch() { (void)hipFree(buf); }
};

template <typename Launch>
static int timed_loop(Launch launch, int warmup, int iters, size_t flush_bytes,
                      double *ms_each, hipStream_t s, bool flush_rmw = false,
                      flush_scratch *shared = NULL) {
    int rc = 0;
    hipEvent_t e0 = NULL, e1 = NULL;
    flush_scratch own;
    flush_scratch *sc = shared ? shared : &own;
    double *scratch = NULL;
    size_t nflush = flush_bytes / sizeof(double);
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    if (nflush) {
        rc = sc->reserve(nflush * sizeof(double));
        if (rc)
            goto fail;
        scratch = sc->buf;
    }
    for (int it = -warmup; it < iters; ++it) {
        if (nflush && flush_rmw)
            hipLaunchKernelGGL(k_flush, dim3(2048), dim3(256), 0, s, scratch,
                               nflush);
        else if (nflush)
            hipLaunchKernelGGL(k_flush_ro, dim3(2048), dim3(256), 0, s,
                               scratch, nflush, scratch);
        HIP_TRY(hipEventRecord(e0, s));
        rc = launch();
        if (rc)
            goto fail;
        HIP_TRY(hipEventRecord(e1, s));
        HIP_TRY(hipEventSynchronize(e1));
        if (it >= 0) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            ms_each[it] = (double)ms;
        }
    }
fail:
    if (e0)
        (void)hipEventDestroy(e0);
    if (e1)
        (void)hipEventDestroy(e1);
    return rc;
}

extern "C" {

int spmv_csr_time(const spmv_csr_dev *A, int kernel,
                  const spmv_launch_opts *opts, const double *d_x, double *d_y,
                  int warmup, int iters, size_t flush_bytes, double *ms_each,
                  void *stream) {
    HANDLE_OK(A); /* before the scratch buffer of the flush is allocated */
    if (iters < 0 || warmup < 0 || (iters && !ms_each))
        return -EINVAL;
    return timed_loop(
        [&]() { return spmv_csr_launch(A, kernel, opts, d_x, d_y, stream); },
        warmup, iters, flush_bytes, ms_each, (hipStream_t)stream,
        opts && (opts->variant & SPMV_VARIANT_FLUSH_RMW));
}

int spmv_hll_time(const spmv_hll_dev *H, int kernel,
                  const spmv_launch_opts *opts, const double *d_x, double *d_y,
                  int warmup, int iters, size_t flush_bytes, double *ms_each,
                  void *stream) {
    HANDLE_OK(H);
    if (iters < 0 || warmup < 0 || (iters && !ms_each))
        return -EINVAL;
    return timed_loop(
        [&]() { return spmv_hll_launch(H, kernel, opts, d_x, d_y, stream); },
        warmup, iters, flush_bytes, ms_each, (hipStream_t)stream,
        opts && (opts->variant & SPMV_VARIANT_FLUSH_RMW));
}

/* ------------------------------------------------------------------ */
/* kernel selection by measurement                                      */
/* ------------------------------------------------------------------ */

static double median_of(std::vector<double> v) {
    std::sort(v.begin(), v.end());
    return v.empty() ? 0.0 : v[v.size() / 2];
}

/* Candidates are timed in the regime they will run in.  A matrix whose
 * working set fits the 256 MiB Infinity Cache would otherwise be tuned on
 * cache hits (config 2, 212 MB: the blocked copy won cached, 0.0479 vs
 * 0.0463 ms for the stream kernel flushed) while an application that touches
 * anything else between two products -- and the benchmark, SURVEY 8d: "MALL
 * flushed between iterations for working sets < 512 MB" -- sees HBM. */
static size_t tune_flush_bytes(int64_t algorithmic_bytes) {
    return algorithmic_bytes < ((int64_t)512 << 20) ? (size_t)1 << 30 : 0;
}

int spmv_hll_autotune(spmv_hll_dev *H, const double *d_x, double *d_y,
                      int allow_panels, int *best_kernel, double *best_ms) {
    HANDLE_OK(H);
    if (!best_kernel)
        return -EINVAL;
    const int cand_cm[2] = {1, 2}, cand_rm[2] = {3, 0};
    const int *cand = H->col_major ? cand_cm : cand_rm;
    int best = -1;
    double bms = 1e300;
    std::vector<double> ms(5);
    const size_t flush = tune_flush_bytes(spmv_hll_algorithmic_bytes(H));
    int best_order = H->order;
    tune_log log;
    char line[200];
    const double t_begin = panels_ops::now_s();
    memset(H->tune_ms, 0, sizeof H->tune_ms);
    /* ONE scratch buffer for the cache flush of every timed loop below (a
     * 1 GiB hipMalloc + memset + hipFree per loop otherwise) */
    flush_scratch scratch;
    int rc = scratch.reserve(flush);
    if (rc)
        return rc;
    int iters = 5, warm = 1;
    {
        /* probe: ONE launch of the first candidate.  Beyond 10 ms (a hack
         * block as wide as a hub row of 10^5 entries: 40-60 ms per launch)
         * every configuration gets a single launch and no warm-up -- the
         * blocked copy is what such a matrix will run anyway */
        spmv_launch_opts o;
        memset(&o, 0, sizeof o);
        o.variant = 1;
        rc = timed_loop(
            [&]() { return spmv_hll_launch(H, cand[0], &o, d_x, d_y, NULL); }, 0,
            1, flush, ms.data(), NULL, false, &scratch);
        if (rc)
            return rc;
        if (ms[0] > 10.0) {
            iters = 1;
            warm = 0;
        }
    }
    for (int k = 0; k < 2; ++k)
        for (int order = 0; order < 3; ++order) { /* the workgroup orders */
            if (!H->col_major && order > 0)
                continue; /* the row-major kernels have one order */
            spmv_launch_opts o;
            memset(&o, 0, sizeof o);
            o.variant = 1 << order; /* bit 0 hardware, 1 ranges, 2 grouped */
            /* 5 launches each -- 1 once a launch has run beyond 10 ms (a hack
             * block as wide as a hub row of 10^5 entries: 40-60 ms per
             * launch, 1.8 s for the six combinations otherwise) */
            rc = timed_loop(
                [&]() { return spmv_hll_launch(H, cand[k], &o, d_x, d_y, NULL); },
                warm, iters, flush, ms.data(), NULL, false, &scratch);
            if (rc)
                return rc;
            double m = median_of(std::vector<double>(ms.begin(), ms.begin() + iters));
            if (m > 10.0)
                iters = 1;
            if (H->tune_ms[cand[k]] == 0.0 || m < H->tune_ms[cand[k]])
                H->tune_ms[cand[k]] = m;
            /* another order has to win by 2 % over hardware order */
            if (m < (order == 0 || best != cand[k] ? bms : 0.98 * bms)) {
                bms = m;
                best = cand[k];
                best_order = order;
            }
        }
    H->order = best_order;
    snprintf(line, sizeof line,
             "direct kernels: %.3f s; slots / nnz = %.3f (padding of the "
             "format), widest hack block %d",
             panels_ops::now_s() - t_begin,
             H->NZ > 0 ? (double)H->slots / (double)H->NZ : 0.0, H->max_width);
    log(line);
    if (allow_panels) {
        const double stream_ms =
            (double)spmv_hll_algorithmic_bytes(H) / 7.0e9; /* at 7 TB/s */
        panels_pool_begin(); /* candidates reuse each other's blocks */
        rc = tune_blocked<spmv_panels, panels_ops>(
            &H->panels, H->M, H->M > 0 ? (double)H->NZ / H->M : 0.0, stream_ms,
            &bms,
            [&](int sched, int tile_rows, spmv_panels **out) {
                return panels_from_hll(H, 0, sched, tile_rows, out);
            },
            [&](double *m) {
                int r = timed_loop(
                    [&]() {
                        return spmv_hll_launch(H, SPMV_HLL_KERNEL_PANELS, NULL,
                                               d_x, d_y, NULL);
                    },
                    1, 5, flush, ms.data(), NULL, false, &scratch);
                *m = median_of(ms);
                return r;
            },
            [&](const char *l) { log(l); }); /* by reference */
        panels_pool_end();
        if (rc < 0)
            return rc;
        if (rc > 0) {
            best = SPMV_HLL_KERNEL_PANELS;
            H->tune_ms[SPMV_HLL_KERNEL_PANELS] = bms;
        }
    }
    snprintf(line, sizeof line, "total %.3f s", panels_ops::now_s() - t_begin);
    log(line);
    free(H->tune_log);
    H->tune_log = log.release();
    *best_kernel = best;
    if (best_ms)
        *best_ms = bms;
    return 0;
}

/*
 * CSR candidates.  Always: the sub-wave kernel (three workgroup orders) and
 * the stream kernel (two).  By the shape of the rows:
 *   wave_row    (1)  mean row length >= 48 (a 64-lane wavefront per row wastes
 *                    its lanes below that);
 *   thread_row  (0)  mean row length < 6 -- the reference's plots show its
 *                    thread-per-row kernel winning on roadNet / amazon
 *                    (cuda_csr.cu:19-31), so the selector
 *                    MEASURES it on such matrices instead of assuming;
 *   block_row   (3)  longest row > 64 x the mean (cuda_csr.cu:96-140: the
 *                    kernel for very long rows; it pays a workgroup per row).
 *                    Both extras are timed with ONE launch first and dropped
 *                    there when that runs beyond 20x the best so far.
 * profiles/r04_autotune_irregular.txt records what wins where.
 */
int spmv_csr_autotune(spmv_csr_dev *A, const double *d_x, double *d_y,
                      int allow_panels, int *best_kernel, double *best_ms) {
    HANDLE_OK(A);
    if (!best_kernel)
        return -EINVAL;
    const double mean = A->M > 0 ? (double)A->NZ / A->M : 0.0;
    const int cand[5] = {4, 2, 1, 0, 3}; /* stream first: it is never the slow one */
    int best = -1;
    double bms = 1e300;
    std::vector<double> ms(5);
    const size_t flush = tune_flush_bytes(spmv_csr_algorithmic_bytes(A));
    tune_log log;
    char line[200];
    const double t_begin = panels_ops::now_s();
    memset(A->tune_ms, 0, sizeof A->tune_ms);
    flush_scratch scratch;
    int rc = scratch.reserve(flush);
    if (rc)
        return rc;
    /* 5 launches per configuration -- 1 once any launch has run beyond 10 ms
     * (a hub row of 10^5 entries under the sub-wave kernel: 46 ms x 3 orders
     * x 6 launches was most of a 1.2 s selector run) */
    int iters_cap = 5;
    auto time_k = [&](int kernel, int variant, int iters, double *m) {
        spmv_launch_opts o;
        memset(&o, 0, sizeof o);
        o.variant = variant;
        iters = std::min(iters, iters_cap);
        int r = timed_loop(
            [&]() { return spmv_csr_launch(A, kernel, &o, d_x, d_y, NULL); }, 1,
            iters, flush, ms.data(), NULL, false, &scratch);
        *m = median_of(std::vector<double>(ms.begin(), ms.begin() + iters));
        if (*m > 10.0)
            iters_cap = 1;
        return r;
    };
    for (int k = 0; k < 5; ++k) {
        if (cand[k] == 1 && mean < 48.0)
            continue; /* a wavefront per row wastes lanes on short rows */
        if (cand[k] == 0 && !(mean < 6.0))
            continue; /* a lane per row: very short rows only */
        if (cand[k] == 3 && !(A->M > 0 && (double)A->max_row_len > 64.0 * mean))
            continue; /* a workgroup per row: a far longer row than the rest */
        double m;
        if (cand[k] == 2) { /* the sub-wave kernel in its three orders */
            static const int bit[3] = {1, 2, 32};
            double mo[3];
            for (int order = 0; order < 3; ++order) {
                rc = time_k(2, bit[order], 5, &mo[order]);
                if (rc)
                    return rc;
            }
            int pick = 0;
            for (int order = 1; order < 3; ++order)
                if (mo[order] < mo[pick])
                    pick = order;
            A->order = pick;
            m = mo[pick];
        } else if (cand[k] == 4) { /* the stream kernel in its two orders */
            double mo[2];
            for (int grp = 0; grp < 2; ++grp) {
                rc = time_k(4, grp ? 32 : 64, 5, &mo[grp]);
                if (rc)
                    return rc;
            }
            A->stream_grouped = mo[1] < mo[0];
            m = mo[A->stream_grouped];
        } else if (cand[k] == 3 || cand[k] == 0) {
            /* one launch first: a workgroup per row over millions of short
             * rows, or one lane walking a row of 10^5 entries, can be orders
             * of magnitude off -- then one sample is the answer */
            rc = time_k(cand[k], 0, 1, &m);
            if (!rc && m < 20.0 * bms)
                rc = time_k(cand[k], 0, cand[k] == 3 ? 2 : 5, &m);
            if (rc)
                return rc;
        } else {
            rc = time_k(cand[k], 0, 5, &m);
            if (rc)
                return rc;
        }
        A->tune_ms[cand[k]] = m;
        if (m < bms) {
            bms = m;
            best = cand[k];
        }
    }
    snprintf(line, sizeof line,
             "direct kernels: %.3f s; mean row %.2f, longest %d; ms: "
             "thread_row %.4f wave_row %.4f subwave_row %.4f block_row %.4f "
             "stream %.4f (0 = not a candidate)",
             panels_ops::now_s() - t_begin, mean, A->max_row_len,
             A->tune_ms[0], A->tune_ms[1], A->tune_ms[2], A->tune_ms[3],
             A->tune_ms[4]);
    log(line);
    if (allow_panels) {
        const double stream_ms = (double)spmv_csr_algorithmic_bytes(A) / 7.0e9;
        panels_pool_begin(); /* candidates reuse each other's blocks */
        rc = tune_blocked<spmv_panels, panels_ops>(
            &A->panels, A->M, mean, stream_ms, &bms,
            [&](int sched, int tile_rows, spmv_panels **out) {
                return panels_from_csr(A, 0, sched, tile_rows, out);
            },
            [&](double *m) {
                int r = timed_loop(
                    [&]() {
                        return spmv_csr_launch(A, SPMV_CSR_KERNEL_PANELS, NULL,
                                               d_x, d_y, NULL);
                    },
                    1, 5, flush, ms.data(), NULL, false, &scratch);
                *m = median_of(ms);
                return r;
            },
            [&](const char *l) { log(l); }); /* by reference */
        panels_pool_end();
        if (rc < 0)
            return rc;
        if (rc > 0) {
            best = SPMV_CSR_KERNEL_PANELS;
            A->tune_ms[SPMV_CSR_KERNEL_PANELS] = bms;
        }
    }
    snprintf(line, sizeof line, "total %.3f s", panels_ops::now_s() - t_begin);
    log(line);
    free(A->tune_log);
    A->tune_log = log.release();
    *best_kernel = best;
    if (best_ms)
        *best_ms = bms;
    return 0;
}

/* per-kernel medians of the last spmv_*_autotune (ms; 0 = not a candidate),
 * kernel ids 0 .. n-1 */
int spmv_csr_tune_times(const spmv_csr_dev *A, double *ms, int n) {
    HANDLE_OK(A);
    if (!ms || n < 0)
        return -EINVAL;
    for (int k = 0; k < n; ++k)
        ms[k] = k < 8 ? A->tune_ms[k] : 0.0;
    return 0;
}

int spmv_hll_tune_times(const spmv_hll_dev *H, double *ms, int n) {
    HANDLE_OK(H);
    if (!ms || n < 0)
        return -EINVAL;
    for (int k = 0; k < n; ++k)
        ms[k] = k < 8 ? H->tune_ms[k] : 0.0;
    return 0;
}

static int copy_log(const char *log, char *buf, size_t len) {
    if (!buf || !len)
        return -EINVAL;
    if (!log)
        return -ENOENT; /* never tuned */
    snprintf(buf, len, "%s", log);
    return 0;
}

/* what the last spmv_*_autotune did, one line per phase with host-clock
 * seconds (build / timing per blocked candidate); -ENOENT before any */
int spmv_csr_tune_log(const spmv_csr_dev *A, char *buf, size_t len) {
    HANDLE_OK(A);
    return copy_log(A->tune_log, buf, len);
}

int spmv_hll_tune_log(const spmv_hll_dev *H, char *buf, size_t len) {
    HANDLE_OK(H);
    return copy_log(H->tune_log, buf, len);
}

/* ------------------------------------------------------------------ */
/* one-shot entry points: the reference's seam (hip_csr.h / hip_hll.h)  */
/* upload -> one event-timed launch -> download -> release              */
/* (reference cuda_csr.cu:210-234, cuda_hll.cu:235-260)                 */
/* ------------------------------------------------------------------ */

/*
 * Opt-in "keep the last upload" behind the seam (spmv_seam_cache, hip_csr.h).
 * The reference's driver calls the seam 27 times per matrix on the SAME A /
 * H_row / H_col and the same x (main.c:258-354), and every call allocates,
 * uploads and frees the whole matrix (cuda_csr.cu:180-205; HLL: per hack
 * block, cuda_hll.cu:161-206): config 2 through this seam is 6.3 ms of wall
 * time around a 0.044 ms kernel.  With the cache on, a call whose matrix is
 * the one uploaded last (same pointers, shape, and a 64-bit fingerprint of
 * the arrays' heads and tails) reuses the device copy, its x / y buffers and
 * -- level 2 -- the uploaded x when its fingerprint is unchanged.  Three
 * slots: CSR, row-major HLL, col-major HLL (what one run of the driver
 * holds).  A different matrix replaces its slot; level 0 releases all.
 */
struct seam_slot {
    const void *host;      /* the caller's struct */
    int device;            /* the HIP device the copy lives on: a call made
                              with another device current is a miss (the
                              uncached path and the reference's seam always
                              use the current device) */
    uint64_t print;        /* fingerprint of the matrix */
    spmv_csr_dev *csr;
    spmv_hll_dev *hll;
    double *d_x, *d_y;
    int M, N;
    const double *x_host;  /* level 2: the x that d_x holds */
    uint64_t x_print;
    bool x_valid;
};
static struct {
    std::mutex mu;
    int level;
    seam_slot slot[3];
    long hits, misses;
} g_seam;

static uint64_t fp_mix(uint64_t h, uint64_t v) {
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xBF58476D1CE4E5B9ull;
    return h ^ (h >> 29);
}

/* head, tail and 64 strided samples of an array of n 4- or 8-byte items */
static uint64_t fp_array(uint64_t h, const void *p, size_t n, size_t item) {
    h = fp_mix(h, (uint64_t)(uintptr_t)p);
    h = fp_mix(h, n);
    if (!p || !n)
        return h;
    const unsigned char *b = (const unsigned char *)p;
    auto at = [&](size_t i) {
        uint64_t v = 0;
        memcpy(&v, b + i * item, item);
        h = fp_mix(h, v);
    };
    const size_t edge = n < 64 ? n : 64;
    for (size_t i = 0; i < edge; ++i)
        at(i);
    for (size_t i = n - edge; i < n; ++i)
        at(i);
    const size_t step = n / 64 ? n / 64 : 1;
    for (size_t i = 0; i < n; i += step)
        at(i);
    return h;
}

/*
 * Level 3: EVERY byte of the arrays, every call -- the one level that cannot
 * return a stale result whatever the caller changed in place.  Four
 * independent multiply-rotate lanes per 1 MiB chunk (instruction-level
 * parallelism), chunks hashed by up to 8 host threads and combined in order.
 * Priced on config 2 (212 MB of IRP / JA / AS + 8 MB of x): see DESIGN.md.
 */
static uint64_t fp_chunk(const unsigned char *b, size_t bytes) {
    uint64_t h[4] = {0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full,
                     0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull};
    size_t i = 0;
    for (; i + 32 <= bytes; i += 32) {
        uint64_t v[4];
        memcpy(v, b + i, 32);
        for (int k = 0; k < 4; ++k) {
            h[k] = (h[k] ^ v[k]) * 0xFF51AFD7ED558CCDull;
            h[k] = (h[k] << 29) | (h[k] >> 35);
        }
    }
    uint64_t tail = 0, out = fp_mix(fp_mix(h[0], h[1]), fp_mix(h[2], h[3]));
    for (; i < bytes; ++i) {
        tail = (tail << 8) | b[i];
        if ((i & 7) == 7) {
            out = fp_mix(out, tail);
            tail = 0;
        }
    }
    return fp_mix(out, tail ^ bytes);
}

static uint64_t fp_full(uint64_t h, const void *p, size_t n, size_t item) {
    h = fp_mix(h, (uint64_t)(uintptr_t)p);
    h = fp_mix(h, n);
    if (!p || !n)
        return h;
    const unsigned char *b = (const unsigned char *)p;
    const size_t bytes = n * item, CH = (size_t)1 << 20;
    const size_t chunks = (bytes + CH - 1) / CH;
    std::vector<uint64_t> part(chunks);
    int T = spmv_host_threads();
    T = T > 8 ? 8 : T;
    if ((size_t)T > chunks / 4)
        T = (int)(chunks / 4); /* at least 4 MiB per thread */
    auto work = [&](size_t c0, size_t c1) {
        for (size_t c = c0; c < c1; ++c)
            part[c] = fp_chunk(b + c * CH, std::min(CH, bytes - c * CH));
    };
    if (T <= 1) {
        work(0, chunks);
    } else {
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t)
            th.emplace_back(work, chunks * t / T, chunks * (t + 1) / T);
        work(0, chunks / T);
        for (std::thread &t : th)
            t.join();
    }
    for (uint64_t v : part)
        h = fp_mix(h, v);
    return h;
}

/* the fingerprint of one array at the cache's current level (g_seam.mu held) */
static uint64_t fp_level(uint64_t h, const void *p, size_t n, size_t item);

/* releases what the slot holds, with the device that owns it current */
static void seam_drop(seam_slot *c) {
    int cur = -1;
    const bool held = c->csr || c->hll || c->d_x || c->d_y;
    if (held && hipGetDevice(&cur) == hipSuccess && cur != c->device)
        (void)hipSetDevice(c->device);
    else
        cur = -1;
    if (c->csr)
        spmv_csr_release(c->csr);
    if (c->hll)
        spmv_hll_release(c->hll);
    (void)hipFree(c->d_x);
    (void)hipFree(c->d_y);
    if (cur >= 0)
        (void)hipSetDevice(cur);
    memset(c, 0, sizeof *c);
}

void spmv_seam_cache(int level) {
    std::lock_guard<std::mutex> g(g_seam.mu);
    const int old = g_seam.level;
    g_seam.level = level < 0 ? 0 : (level > 3 ? 3 : level);
    /* the fingerprints of levels 1 / 2 (samples) and 3 (every byte) are
     * different functions: what is held was keyed under the old one */
    if (g_seam.level == 0 || (old != 0 && (old == 3) != (g_seam.level == 3)))
        for (seam_slot &c : g_seam.slot)
            seam_drop(&c);
}

static uint64_t fp_level(uint64_t h, const void *p, size_t n, size_t item) {
    return g_seam.level >= 3 ? fp_full(h, p, n, item) : fp_array(h, p, n, item);
}

/* Targeted invalidate: `host` = a matrix struct (sparse_csr* / sparse_hll*)
 * handed to the seam earlier -> its device copy is dropped (the next call
 * uploads again); `host` = an x vector -> the next call uploads x again;
 * NULL -> everything held is dropped, the level stays.  For a caller that
 * edits a matrix or x IN PLACE and runs level 1 / 2 (sampled fingerprints).
 * Returns how many slots it touched. */
int spmv_seam_cache_invalidate(const void *host) {
    std::lock_guard<std::mutex> g(g_seam.mu);
    int n = 0;
    for (seam_slot &c : g_seam.slot) {
        if (!(c.csr || c.hll))
            continue;
        if (!host || c.host == host) {
            seam_drop(&c);
            ++n;
        } else if (c.x_host == host && c.x_valid) {
            c.x_valid = false;
            ++n;
        }
    }
    return n;
}

int spmv_seam_cache_stats(long *hits, long *misses) {
    std::lock_guard<std::mutex> g(g_seam.mu);
    if (hits)
        *hits = g_seam.hits;
    if (misses)
        *misses = g_seam.misses;
    int held = 0;
    for (const seam_slot &c : g_seam.slot)
        held += c.csr || c.hll;
    return held;
}

static int one_shot_vectors(int M, int N, const double *x, double **d_x,
                            double **d_y) {
    int rc = 0;
    *d_x = *d_y = NULL;
    HIP_TRY(hipMalloc((void **)d_x, std::max<size_t>(N, 1) * sizeof(double)));
    HIP_TRY(hipMalloc((void **)d_y, std::max<size_t>(M, 1) * sizeof(double)));
    if (N > 0)
        HIP_TRY(hipMemcpy(*d_x, x, (size_t)N * sizeof(double),
                          hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(*d_y, 0, std::max<size_t>(M, 1) * sizeof(double)));
fail:
    return rc;
}

/* the cached form of upload + vectors: slot `which` (0 CSR, 1 HLL row-major,
 * 2 HLL col-major); `upload` builds the device copy on a miss */
static int seam_acquire(int which, const void *host, uint64_t print, int M,
                        int N, const double *x,
                        const std::function<int(seam_slot *)> &upload,
                        seam_slot **out) {
    int rc = 0;
    seam_slot *c = &g_seam.slot[which];
    int dev = 0;
    HIP_RET(hipGetDevice(&dev));
    const bool hit = (c->csr || c->hll) && c->host == host && c->print == print &&
                     c->M == M && c->N == N && c->device == dev;
    if (!hit) {
        seam_drop(c);
        ++g_seam.misses;
        c->device = dev;
        rc = upload(c);
        if (rc) {
            seam_drop(c);
            return rc;
        }
        c->host = host;
        c->print = print;
        c->M = M;
        c->N = N;
        HIP_TRY(hipMalloc((void **)&c->d_x,
                          std::max<size_t>(N, 1) * sizeof(double)));
        HIP_TRY(hipMalloc((void **)&c->d_y,
                          std::max<size_t>(M, 1) * sizeof(double)));
    } else {
        ++g_seam.hits;
    }
    {
        const uint64_t xp = g_seam.level >= 2
                                ? fp_level(0x78, x, (size_t)N, sizeof(double))
                                : 0;
        if (!(g_seam.level >= 2 && c->x_valid && c->x_host == x &&
              c->x_print == xp)) {
            if (N > 0)
                HIP_TRY(hipMemcpy(c->d_x, x, (size_t)N * sizeof(double),
                                  hipMemcpyHostToDevice));
            c->x_host = x;
            c->x_print = xp;
            c->x_valid = g_seam.level >= 2;
        }
    }
    HIP_TRY(hipMemsetAsync(c->d_y, 0, std::max<size_t>(M, 1) * sizeof(double),
                           0));
    *out = c;
    return 0;
fail:
    seam_drop(c);
    return rc;
}

static double csr_one_shot(const sparse_csr *A, const double *x, double *y,
                           void *arg, int kernel) {
    if (!A || !x || !y)
        return -EINVAL;
    const spmv_launch_opts *opts = (const spmv_launch_opts *)arg;
    double ms = 0.0;
    int rc = 0;
    {
        std::unique_lock<std::mutex> g(g_seam.mu);
        if (g_seam.level > 0) {
            const double t0s = panels_ops::now_s();
            uint64_t fp = fp_mix(fp_mix(fp_mix(0x637372, (uint64_t)A->M),
                                        (uint64_t)A->N), (uint64_t)A->NZ);
            fp = fp_level(fp, A->IRP, (size_t)A->M + 1, sizeof(int));
            fp = fp_level(fp, A->JA, (size_t)A->NZ, sizeof(int));
            fp = fp_level(fp, A->AS, (size_t)A->NZ, sizeof(double));
            seam_slot *c = NULL;
            const double t1 = panels_ops::now_s();
            rc = seam_acquire(0, A, fp, A->M, A->N, x,
                              [&](seam_slot *s) {
                                  return spmv_csr_upload(A, &s->csr);
                              },
                              &c);
            const double t2 = panels_ops::now_s();
            if (!rc)
                rc = spmv_csr_time(c->csr, kernel, opts, c->d_x, c->d_y, 0, 1, 0,
                                   &ms, NULL);
            const double t3 = panels_ops::now_s();
            if (!rc && A->M > 0)
                rc = spmv_copy_d2h(y, c->d_y, (size_t)A->M * sizeof(double));
            if (live().debug) /* where a cached call spends its host time */
                fprintf(stderr,
                        "spmv seam cache (csr): fingerprint %.3f ms, acquire "
                        "%.3f, launch + sync %.3f, y download %.3f\n",
                        (t1 - t0s) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3,
                        (panels_ops::now_s() - t3) * 1e3);
            return rc ? (double)rc : ms;
        }
    }
    spmv_csr_dev *d = NULL;
    double *d_x = NULL, *d_y = NULL;
    rc = spmv_csr_upload(A, &d);
    if (rc)
        return rc;
    rc = one_shot_vectors(A->M, A->N, x, &d_x, &d_y);
    if (!rc)
        rc = spmv_csr_time(d, kernel, opts, d_x, d_y, 0, 1, 0, &ms, NULL);
    if (!rc && A->M > 0)
        rc = spmv_copy_d2h(y, d_y, (size_t)A->M * sizeof(double));
    (void)hipFree(d_x);
    (void)hipFree(d_y);
    spmv_csr_release(d);
    return rc ? (double)rc : ms;
}

static double hll_one_shot(const sparse_hll *H, const double *x, double *y,
                           void *arg, int kernel, int col_major) {
    if (!H || !x || !y)
        return -EINVAL;
    const spmv_launch_opts *opts = (const spmv_launch_opts *)arg;
    double ms = 0.0;
    int rc = 0;
    {
        std::unique_lock<std::mutex> g(g_seam.mu);
        if (g_seam.level > 0) {
            uint64_t fp = fp_mix(fp_mix(fp_mix(0x686c6c, (uint64_t)H->M),
                                        (uint64_t)H->N), (uint64_t)H->NZ);
            fp = fp_mix(fp, (uint64_t)H->num_blocks);
            fp = fp_mix(fp, (uint64_t)(uintptr_t)H->blocks);
            /* first, middle and last hack block: shape + array fingerprints
             * (level 3: EVERY block, every byte) */
            const int pick[3] = {0, H->num_blocks / 2, H->num_blocks - 1};
            const int npick = g_seam.level >= 3 ? H->num_blocks
                                                : (H->num_blocks > 0 ? 3 : 0);
            for (int k = 0; k < npick; ++k) {
                const ellpack_block *b =
                    &H->blocks[g_seam.level >= 3 ? k : pick[k]];
                const size_t n = (size_t)b->M * (size_t)b->max_NZ;
                fp = fp_mix(fp, ((uint64_t)b->M << 32) | (uint32_t)b->max_NZ);
                if (g_seam.level >= 3) { /* small arrays: no thread spawn */
                    fp = fp_mix(fp, fp_chunk((const unsigned char *)b->JA,
                                             n * sizeof(int)));
                    fp = fp_mix(fp, fp_chunk((const unsigned char *)b->AS,
                                             n * sizeof(double)));
                } else {
                    fp = fp_array(fp, b->JA, n, sizeof(int));
                    fp = fp_array(fp, b->AS, n, sizeof(double));
                }
            }
            seam_slot *c = NULL;
            rc = seam_acquire(col_major ? 2 : 1, H, fp, H->M, H->N, x,
                              [&](seam_slot *s) {
                                  return spmv_hll_upload(H, col_major, &s->hll);
                              },
                              &c);
            if (!rc)
                rc = spmv_hll_time(c->hll, kernel, opts, c->d_x, c->d_y, 0, 1, 0,
                                   &ms, NULL);
            if (!rc && H->M > 0)
                rc = spmv_copy_d2h(y, c->d_y, (size_t)H->M * sizeof(double));
            return rc ? (double)rc : ms;
        }
    }
    spmv_hll_dev *d = NULL;
    double *d_x = NULL, *d_y = NULL;
    rc = spmv_hll_upload(H, col_major, &d);
    if (rc)
        return rc;
    rc = one_shot_vectors(H->M, H->N, x, &d_x, &d_y);
    if (!rc)
        rc = spmv_hll_time(d, kernel, opts, d_x, d_y, 0, 1, 0, &ms, NULL);
    if (!rc && H->M > 0)
        rc = spmv_copy_d2h(y, d_y, (size_t)H->M * sizeof(double));
    (void)hipFree(d_x);
    (void)hipFree(d_y);
    spmv_hll_release(d);
    return rc ? (double)rc : ms;
}

double csr_spmv_hip_thread_row(const sparse_csr *A, const double *x, double *y,
                               void *arg) {
    return csr_one_shot(A, x, y, arg, 0);
}
double csr_spmv_hip_wave_row(const sparse_csr *A, const double *x, double *y,
                             void *arg) {
    return csr_one_shot(A, x, y, arg, 1);
}
double csr_spmv_hip_subwave_row(const sparse_csr *A, const double *x,
                                double *y, void *arg) {
    return csr_one_shot(A, x, y, arg, 2);
}
double csr_spmv_hip_block_row(const sparse_csr *A, const double *x, double *y,
                              void *arg) {
    return csr_one_shot(A, x, y, arg, 3);
}
double csr_spmv_hip_stream(const sparse_csr *A, const double *x, double *y,
                           void *arg) {
    return csr_one_shot(A, x, y, arg, 4);
}

double hll_spmv_hip_threads_row_major(const sparse_hll *H, const double *x,
                                      double *y, void *arg) {
    return hll_one_shot(H, x, y, arg, 0, 0);
}
double hll_spmv_hip_threads_col_major(const sparse_hll *H, const double *x,
                                      double *y, void *arg) {
    return hll_one_shot(H, x, y, arg, 1, 1);
}
double hll_spmv_hip_wave_block(const sparse_hll *H, const double *x, double *y,
                               void *arg) {
    return hll_one_shot(H, x, y, arg, 2, 1);
}
double hll_spmv_hip_subwave_row(const sparse_hll *H, const double *x,
                                double *y, void *arg) {
    return hll_one_shot(H, x, y, arg, 3, 0);
}

/* ------------------------------------------------------------------ */
/* the reference's own symbol names (include/spmv_ref_abi.h)            */
/* ------------------------------------------------------------------ */
void set_csr_warps_per_block(int w) { set_csr_waves_per_block(w); }
void set_hll_warps_per_block(int w) { set_hll_waves_per_block(w); }

double csr_spmv_cuda_thread_row(const sparse_csr *A, const double *x,
                                double *y, void *unused) {
    (void)unused;
    return csr_one_shot(A, x, y, NULL, 0);
}
double csr_spmv_cuda_warp_row(const sparse_csr *A, const double *x, double *y,
                              void *unused) {
    (void)unused;
    return csr_one_shot(A, x, y, NULL, 1);
}
double csr_spmv_cuda_halfwarp_row(const sparse_csr *A, const double *x,
                                  double *y, void *unused) {
    (void)unused;
    return csr_one_shot(A, x, y, NULL, 2);
}
double csr_spmv_cuda_block_row(const sparse_csr *A, const double *x, double *y,
                               void *unused) {
    (void)unused;
    return csr_one_shot(A, x, y, NULL, 3);
}
double csr_spmv_cuda_halfwarp_row_text(const sparse_csr *A, const double *x,
                                       double *y, void *unused) {
    (void)unused;
    return csr_one_shot(A, x, y, NULL, 4);
}
double hll_spmv_cuda_threads_row_major(const sparse_hll *H, const double *x,
                                       double *y, void *unused) {
    (void)unused;
    return hll_one_shot(H, x, y, NULL, 0, 0);
}
double hll_spmv_cuda_threads_col_major(const sparse_hll *H, const double *x,
                                       double *y, void *unused) {
    (void)unused;
    return hll_one_shot(H, x, y, NULL, 1, 1);
}
double hll_spmv_cuda_warp_block(const sparse_hll *H, const double *x,
                                double *y, void *unused) {
    (void)unused;
    return hll_one_shot(H, x, y, NULL, 2, 1);
}
double hll_spmv_cuda_halfwarp_row(const sparse_hll *H, const double *x,
                                  double *y, void *unused) {
    (void)unused;
    return hll_one_shot(H, x, y, NULL, 3, 0);
}

} /* extern "C" */

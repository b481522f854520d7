/*
 * mgpu.hip -- single-process multi-GPU SpMV for the C driver: row-range
 * partition over the GPUs of one node + RCCL all-gather of the y fragments
 * over xGMI (API: include/spmv_mgpu.h).  New; the reference is single-GPU.
 *
 * One host thread drives all devices: per device a stream, a shard handle
 * (CSR or col-major HLL of the local rows with GLOBAL column indices), the
 * full-length x and the full-length y.  A step launches every shard's kernel
 * (asynchronous), then one grouped in-place ncclAllGather
 * (sendbuff = y + rank*rows) so every device ends with the whole y.
 * bench.py does the same with one process per GPU through torch.distributed;
 * this is the path of `spmv_scpa_amd -g N`.
 */
#include <mutex>
#include <thread>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <vector>

#include "err.h"
#include "hip_common.h"
#include "spmv_mgpu.h"

struct spmv_mgpu {
    int n;
    int rows_per_gpu; /* equal shards, multiple of 32 (last one padded) */
    int M, N;         /* global shape */
    int is_hll;
    std::vector<int> dev;
    std::vector<hipStream_t> stream;
    std::vector<ncclComm_t> comm;
    std::vector<spmv_csr_dev *> csr; /* [device]: the shard -- with logical */
    std::vector<spmv_hll_dev *> hll; /* shards, logical shard 0 of the device */
    /* LOGICAL SHARDS (spmv_mgpu_set_logical_shards): a device's rows as L
     * matrices of rows / L rows each (global columns, local rows), so that a
     * kernel that runs whole matrices only -- the blocked path -- still
     * overlaps: logical shard c of every device is all-gathered on the second
     * stream while shard c+1 computes (the staged pipeline, chunk = shard).
     * Shards 1 .. L-1 of device r live at [r * (L - 1) + c - 1]. */
    int L, want_L, reserve_cus;
    /* exchange ENGINE: 0 = RCCL collectives (kernels on the CUs), 1 = the
     * copy engines -- every device PUSHES its fragment into its peers' y with
     * hipMemcpyAsync over xGMI (SDMA, peer access enabled at create): no
     * kernel competes with the SpMV for compute units, which is what the
     * persistent sweep launch wants, and with logical shards the pushes of
     * shard c run under the kernel of c + 1 without any staging buffer.
     * Rehearsal handles always run this engine (same-device copies). */
    int engine;
    std::vector<spmv_csr_dev *> xcsr;
    std::vector<spmv_hll_dev *> xhll;
    std::vector<double *> x, y;
    /* overlapped exchange (spmv_mgpu_set_exchange): the shard's rows in
     * `chunks` equal pieces; chunk c of every device is written into the
     * chunk-major staging buffer stage[c][device][:] and all-gathered in
     * place on a second stream while the kernel of chunk c+1 runs; one
     * strided copy puts y back into row order */
    int chunks, force_exchange;
    int even; /* every shard holds exactly rows_per_gpu rows (no padded last
                 shard): the chunk launches of the staged pipeline need it */
    /* row ranges: device r owns rows [start[r], start[r+1]).  Even partition:
     * start[r] = r * rows_per_gpu (y is padded to rows_per_gpu * n so that one
     * in-place ncclAllGather moves it).  nnz-balanced partition
     * (partition_rows_nnz_aligned, csr.h): the ranges differ -- `ragged` --
     * y holds exactly M rows and the fragments travel by `xchg`:
     * SPMV_MGPU_XCHG_P2P / _BCAST / _PADDED (spmv_mgpu.h) */
    std::vector<int> start;
    int ragged, xchg, longest;
    /* REHEARSAL handle (spmv_mgpu_create_rehearsal): n logical devices on the
     * visible card(s), no communicator (RCCL refuses two ranks on one device);
     * the exchange is device-to-device copies between the logical devices'
     * y buffers.  Tests the partition / offsets / empty-range logic on a
     * 1-GPU box; its timings mean nothing. */
    int loopback;
    std::vector<double *> pad; /* _PADDED: [n][longest] staging per device */
    std::vector<hipStream_t> xstream;
    std::vector<double *> stage;
    std::vector<hipEvent_t> ev_k, ev_x; /* [device * MG_MAX_CHUNKS + chunk] */
};
#define MG_MAX_CHUNKS 16

/* live multi-GPU handles: a second destroy of one handle is ignored (same
 * contract as spmv_*_release, engine.hip); never destroyed objects */
static std::mutex &mg_mu(void) {
    static std::mutex *m = new std::mutex();
    return *m;
}
static std::vector<const spmv_mgpu *> &mg_live(void) {
    static std::vector<const spmv_mgpu *> *v =
        new std::vector<const spmv_mgpu *>();
    return *v;
}
static bool mg_take(const spmv_mgpu *g) {
    std::lock_guard<std::mutex> lk(mg_mu());
    std::vector<const spmv_mgpu *> &v = mg_live();
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i] == g) {
            v.erase(v.begin() + (long)i);
            return true;
        }
    return false;
}

static bool mg_has(const spmv_mgpu *g) {
    std::lock_guard<std::mutex> lk(mg_mu());
    const std::vector<const spmv_mgpu *> &v = mg_live();
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i] == g)
            return true;
    return false;
}
/* first statement of every entry point: NULL -> -EINVAL, a destroyed (or
 * never created) handle -> -EBADF, before it is dereferenced */
#define MG_OK(g)                                                              \
    do {                                                                      \
        if (!(g))                                                             \
            return -EINVAL;                                                   \
        if (!mg_has(g))                                                       \
            return -EBADF;                                                    \
    } while (0)

static int nccl_errno(ncclResult_t r) {
    return r == ncclSuccess ? 0 : (r == ncclSystemError ? -EIO : -EINVAL);
}

#define NCCL_TRY(call)                                                        \
    do {                                                                      \
        ncclResult_t r_ = (call);                                             \
        if (r_ != ncclSuccess) {                                              \
            rc = nccl_errno(r_);                                              \
            goto fail;                                                        \
        }                                                                     \
    } while (0)

/* Every entry point walks the devices with hipSetDevice; a caller that
 * shares the process (torch, another library) must find ITS current device
 * unchanged afterwards. */
struct device_guard {
    int saved;
    bool ok;
    device_guard() : saved(0), ok(hipGetDevice(&saved) == hipSuccess) {}
    ~device_guard() {
        if (ok)
            (void)hipSetDevice(saved);
    }
};

/* logical shard c of device r (c = 0: the device's own slot) */
static spmv_csr_dev *&csr_at(spmv_mgpu *g, int r, int c) {
    return c == 0 ? g->csr[(size_t)r] : g->xcsr[(size_t)r * (g->L - 1) + c - 1];
}
static spmv_hll_dev *&hll_at(spmv_mgpu *g, int r, int c) {
    return c == 0 ? g->hll[(size_t)r] : g->xhll[(size_t)r * (g->L - 1) + c - 1];
}

/* shards and vectors of a previous load / generate on this handle */
static void drop_shards(spmv_mgpu *g) {
    for (int r = 0; r < g->n; ++r) {
        (void)hipSetDevice(g->dev[r]);
        for (int c = 1; c < g->L; ++c) {
            if (csr_at(g, r, c))
                spmv_csr_release(csr_at(g, r, c));
            if (hll_at(g, r, c))
                spmv_hll_release(hll_at(g, r, c));
        }
        if (g->csr[r])
            spmv_csr_release(g->csr[r]);
        if (g->hll[r])
            spmv_hll_release(g->hll[r]);
        (void)hipFree(g->x[r]);
        (void)hipFree(g->y[r]);
        (void)hipFree(g->stage[r]);
        (void)hipFree(g->pad[r]);
        g->csr[r] = NULL;
        g->hll[r] = NULL;
        g->x[r] = g->y[r] = g->stage[r] = g->pad[r] = NULL;
    }
    g->rows_per_gpu = g->M = g->N = 0;
    g->ragged = g->longest = 0;
    g->start.assign((size_t)g->n + 1, 0);
    g->xcsr.clear();
    g->xhll.clear();
    g->L = 1;
}

/* the logical shards asked for take effect now (a load / generate): needs the
 * even partition with rows per device = L shards of whole hack blocks */
static int adopt_logical_shards(spmv_mgpu *g) {
    g->L = 1;
    if (g->want_L <= 1)
        return 0;
    if (g->ragged || !g->even || g->rows_per_gpu % (g->want_L * HACK_SIZE))
        return -EINVAL;
    g->L = g->want_L;
    g->xcsr.assign((size_t)g->n * (g->L - 1), NULL);
    g->xhll.assign((size_t)g->n * (g->L - 1), NULL);
    return 0;
}

/* a load / generate has succeeded on this handle (vectors exist) */
static bool mg_loaded(const spmv_mgpu *g) {
    if (!g->y[0] || !g->x[0])
        return false;
    for (int r = 0; r < g->n; ++r)
        if (g->csr[(size_t)r] || g->hll[(size_t)r])
            return true;
    return false;
}

extern "C" {

void spmv_mgpu_destroy(spmv_mgpu *g) {
    if (!g || !mg_take(g))
        return; /* NULL or destroyed before: ignored */
    device_guard keep;
    drop_shards(g);
    for (int r = 0; r < g->n; ++r) {
        (void)hipSetDevice(g->dev[r]);
        if (g->comm[r])
            ncclCommDestroy(g->comm[r]);
        if (g->stream[r])
            (void)hipStreamDestroy(g->stream[r]);
        if (g->xstream[r])
            (void)hipStreamDestroy(g->xstream[r]);
        for (int c = 0; c < MG_MAX_CHUNKS; ++c) {
            if (g->ev_k[(size_t)r * MG_MAX_CHUNKS + c])
                (void)hipEventDestroy(g->ev_k[(size_t)r * MG_MAX_CHUNKS + c]);
            if (g->ev_x[(size_t)r * MG_MAX_CHUNKS + c])
                (void)hipEventDestroy(g->ev_x[(size_t)r * MG_MAX_CHUNKS + c]);
        }
    }
    delete g;
}

static int create(int ngpus, int loopback, spmv_mgpu **out);

int spmv_mgpu_create(int ngpus, spmv_mgpu **out) {
    return create(ngpus, 0, out);
}

int spmv_mgpu_create_rehearsal(int nlogical, spmv_mgpu **out) {
    return create(nlogical, 1, out);
}

static int create(int ngpus, int loopback, spmv_mgpu **out) {
    if (!out || ngpus < 1 || ngpus > 64)
        return -EINVAL;
    *out = NULL;
    const int have = spmv_device_count();
    if (have < (loopback ? 1 : ngpus))
        return -ENODEV;
    int rc = 0;
    device_guard keep;
    spmv_mgpu *g = new spmv_mgpu();
    {
        std::lock_guard<std::mutex> lk(mg_mu());
        mg_live().push_back(g);
    }
    g->n = ngpus;
    g->rows_per_gpu = g->M = g->N = g->is_hll = 0;
    g->dev.resize(ngpus);
    g->stream.assign(ngpus, NULL);
    g->comm.assign(ngpus, NULL);
    g->csr.assign(ngpus, NULL);
    g->hll.assign(ngpus, NULL);
    g->x.assign(ngpus, NULL);
    g->y.assign(ngpus, NULL);
    g->stage.assign(ngpus, NULL);
    g->pad.assign(ngpus, NULL);
    g->start.assign((size_t)ngpus + 1, 0);
    g->ragged = g->longest = 0;
    g->xchg = SPMV_MGPU_XCHG_P2P;
    g->xstream.assign(ngpus, NULL);
    g->ev_k.assign((size_t)ngpus * MG_MAX_CHUNKS, NULL);
    g->ev_x.assign((size_t)ngpus * MG_MAX_CHUNKS, NULL);
    g->chunks = 1;
    g->force_exchange = 0;
    g->even = 0;
    g->L = g->want_L = 1;
    g->reserve_cus = 0;
    g->engine = loopback ? 1 : 0;
    g->loopback = loopback;
    for (int r = 0; r < ngpus; ++r)
        g->dev[r] = loopback ? r % have : r;
    if (!loopback)
        NCCL_TRY(ncclCommInitAll(g->comm.data(), ngpus, g->dev.data()));
    for (int r = 0; r < ngpus && !loopback; ++r) {
        /* peer access for the copy engine (and for RCCL's own P2P paths);
         * "already enabled" is not an error, "not supported" leaves the
         * copies to the runtime's staging */
        HIP_TRY(hipSetDevice(g->dev[r]));
        for (int p = 0; p < ngpus; ++p) {
            int can = 0;
            if (p != r &&
                hipDeviceCanAccessPeer(&can, g->dev[r], g->dev[p]) == hipSuccess &&
                can)
                (void)hipDeviceEnablePeerAccess(g->dev[p], 0);
        }
        (void)hipGetLastError();
    }
    for (int r = 0; r < ngpus; ++r) {
        HIP_TRY(hipSetDevice(g->dev[r]));
        HIP_TRY(hipStreamCreate(&g->stream[r]));
        HIP_TRY(hipStreamCreate(&g->xstream[r]));
        for (int c = 0; c < MG_MAX_CHUNKS; ++c) {
            HIP_TRY(hipEventCreateWithFlags(
                &g->ev_k[(size_t)r * MG_MAX_CHUNKS + c], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(
                &g->ev_x[(size_t)r * MG_MAX_CHUNKS + c], hipEventDisableTiming));
        }
    }
    *out = g;
    return 0;
fail:
    spmv_mgpu_destroy(g);
    return rc;
}

/* y of an even partition is padded to rows_per_gpu * n; a ragged one holds
 * exactly M rows */
static size_t y_len(const spmv_mgpu *g) {
    return g->ragged ? (size_t)g->M : (size_t)g->rows_per_gpu * g->n;
}

static int alloc_pad(spmv_mgpu *g) {
    int rc = 0;
    for (int r = 0; r < g->n; ++r) {
        HIP_TRY(hipSetDevice(g->dev[r]));
        (void)hipFree(g->pad[r]);
        g->pad[r] = NULL;
        if (g->ragged && g->xchg == SPMV_MGPU_XCHG_PADDED && g->longest > 0)
            HIP_TRY(hipMalloc((void **)&g->pad[r],
                              (size_t)g->longest * g->n * sizeof(double)));
    }
fail:
    return rc;
}

static int alloc_vectors(spmv_mgpu *g) {
    int rc = 0;
    const size_t ny = y_len(g);
    for (int r = 0; r < g->n; ++r) {
        HIP_TRY(hipSetDevice(g->dev[r]));
        HIP_TRY(hipMalloc((void **)&g->x[r],
                          (size_t)(g->N > 0 ? g->N : 1) * sizeof(double)));
        HIP_TRY(hipMalloc((void **)&g->y[r], (ny ? ny : 1) * sizeof(double)));
        HIP_TRY(hipMemset(g->y[r], 0, (ny ? ny : 1) * sizeof(double)));
        if ((g->chunks > 1 || g->L > 1) && ny > 0 && !g->ragged)
            /* a reload keeps the exchange setting */
            HIP_TRY(hipMalloc((void **)&g->stage[r], ny * sizeof(double)));
    }
    rc = alloc_pad(g);
fail:
    return rc;
}

/* take over a partition: start[], the longest range, even / ragged */
static void set_ranges(spmv_mgpu *g, const int *starts, int M, int by_nnz) {
    g->start.assign(starts, starts + g->n + 1);
    g->longest = 0;
    for (int r = 0; r < g->n; ++r)
        if (starts[r + 1] - starts[r] > g->longest)
            g->longest = starts[r + 1] - starts[r];
    g->rows_per_gpu = g->n == 1 ? M : g->longest;
    /* an nnz partition is handled as ragged even when its ranges happen to be
     * equal (a uniform matrix; one device): one code path per partition kind */
    g->ragged = by_nnz != 0;
    for (int r = 0; r <= g->n; ++r) {
        const long long ev = (long long)r * g->rows_per_gpu;
        if (starts[r] != (int)(ev < M ? ev : M))
            g->ragged = 1;
    }
    g->even = !g->ragged && (long long)g->rows_per_gpu * g->n == (long long)M;
}

/* shard a host matrix: contiguous row ranges, boundaries multiples of 32;
 * partition: SPMV_MGPU_PART_EVEN (equal row counts) or _NNZ (near-equal entry
 * counts, partition_rows_nnz_aligned) */
int spmv_mgpu_load_csr_part(spmv_mgpu *g, const sparse_csr *A, int as_hll,
                            int partition) {
    MG_OK(g);
    if (!g || !A || (partition != SPMV_MGPU_PART_EVEN &&
                     partition != SPMV_MGPU_PART_NNZ))
        return -EINVAL;
    int *starts = partition == SPMV_MGPU_PART_NNZ
                      ? partition_rows_nnz_aligned(A->IRP, A->M, g->n, HACK_SIZE)
                      : partition_rows_even(A->M, g->n, HACK_SIZE);
    if (IS_ERR(starts))
        return PTR_ERR(starts);
    int rc = 0;
    device_guard keep;
    drop_shards(g); /* a handle can be loaded again: nothing leaks */
    g->M = A->M;
    g->N = A->N;
    g->is_hll = as_hll != 0;
    set_ranges(g, starts, A->M, partition == SPMV_MGPU_PART_NNZ);
    rc = adopt_logical_shards(g);
    for (int r = 0; r < g->n && !rc; ++r) {
        HIP_TRY(hipSetDevice(g->dev[r]));
        if (starts[r + 1] == starts[r] && g->n > 1)
            continue; /* more devices than hack blocks: nothing to run here */
        const int per = (starts[r + 1] - starts[r]) / g->L;
        for (int c = 0; c < g->L && !rc; ++c) {
            const int a = starts[r] + c * per;
            sparse_csr *S =
                csr_row_slice(A, a, c == g->L - 1 ? starts[r + 1] : a + per);
            if (IS_ERR(S)) {
                rc = PTR_ERR(S);
                break;
            }
            rc = spmv_csr_upload(S, &csr_at(g, r, c));
            if (!rc && as_hll) {
                rc = spmv_hll_from_csr(csr_at(g, r, c), 1, &hll_at(g, r, c));
                spmv_csr_release(csr_at(g, r, c));
                csr_at(g, r, c) = NULL;
            }
            csr_free(S);
        }
    }
    if (!rc)
        rc = alloc_vectors(g);
fail:
    free(starts);
    if (rc) /* nothing half-loaded stays behind: M, N, the ranges and whatever
               shards were built go; the next step finds an EMPTY handle and
               says -EINVAL instead of handing NULL vectors to a collective */
        drop_shards(g);
    return rc;
}

int spmv_mgpu_load_csr(spmv_mgpu *g, const sparse_csr *A, int as_hll) {
    return spmv_mgpu_load_csr_part(g, A, as_hll, SPMV_MGPU_PART_EVEN);
}

/* every device generates its own shard (spmv_synth.h), weak scaling: the
 * (rows_per_gpu * ngpus)-square matrix cut evenly or -- families with ragged
 * rows -- by entries (partition_synth_rows_nnz: from the row lengths alone) */
int spmv_mgpu_generate_part(spmv_mgpu *g, int kind, int rows_per_gpu, int K,
                            int64_t W, uint64_t seed, int as_hll,
                            int partition) {
    MG_OK(g);
    if (!g || rows_per_gpu < 0 || rows_per_gpu % HACK_SIZE ||
        (long long)rows_per_gpu * g->n > INT32_MAX ||
        (partition != SPMV_MGPU_PART_EVEN && partition != SPMV_MGPU_PART_NNZ))
        return -EINVAL;
    const int M = rows_per_gpu * g->n;
    int *starts = partition == SPMV_MGPU_PART_NNZ && M > 0
                      ? partition_synth_rows_nnz(kind, M, M, K, W, seed, g->n,
                                                 HACK_SIZE)
                      : partition_rows_even(M, g->n, HACK_SIZE);
    if (IS_ERR(starts))
        return PTR_ERR(starts);
    int rc = 0;
    device_guard keep;
    drop_shards(g);
    g->M = g->N = M;
    g->is_hll = as_hll != 0;
    set_ranges(g, starts, M, partition == SPMV_MGPU_PART_NNZ);
    rc = adopt_logical_shards(g);
    for (int r = 0; r < g->n && !rc; ++r) {
        HIP_TRY(hipSetDevice(g->dev[r]));
        if (starts[r + 1] == starts[r] && g->n > 1)
            continue;
        const int per = (starts[r + 1] - starts[r]) / g->L;
        for (int c = 0; c < g->L && !rc; ++c) {
            rc = spmv_csr_generate(kind, per, g->N, K, W,
                                   (int64_t)starts[r] + (int64_t)c * per, seed,
                                   &csr_at(g, r, c));
            if (!rc && as_hll) {
                rc = spmv_hll_from_csr(csr_at(g, r, c), 1, &hll_at(g, r, c));
                spmv_csr_release(csr_at(g, r, c));
                csr_at(g, r, c) = NULL;
            }
        }
    }
    if (!rc)
        rc = alloc_vectors(g);
fail:
    free(starts);
    if (rc) /* nothing half-loaded stays behind: M, N, the ranges and whatever
               shards were built go; the next step finds an EMPTY handle and
               says -EINVAL instead of handing NULL vectors to a collective */
        drop_shards(g);
    return rc;
}

int spmv_mgpu_generate(spmv_mgpu *g, int kind, int rows_per_gpu, int K,
                       int64_t W, uint64_t seed, int as_hll) {
    return spmv_mgpu_generate_part(g, kind, rows_per_gpu, K, W, seed, as_hll,
                                   SPMV_MGPU_PART_EVEN);
}

int spmv_mgpu_set_x(spmv_mgpu *g, const double *x_host) {
    MG_OK(g);
    if (!g || !x_host)
        return -EINVAL;
    if (!mg_loaded(g))
        return -EINVAL;
    int rc = 0;
    device_guard keep;
    for (int r = 0; r < g->n; ++r) {
        HIP_TRY(hipSetDevice(g->dev[r]));
        HIP_TRY(hipMemcpy(g->x[r], x_host, (size_t)g->N * sizeof(double),
                          hipMemcpyHostToDevice));
    }
fail:
    return rc;
}

int spmv_mgpu_fill_x(spmv_mgpu *g, uint64_t seed) {
    MG_OK(g);
    if (!g)
        return -EINVAL;
    if (!mg_loaded(g))
        return -EINVAL;
    int rc = 0;
    device_guard keep;
    for (int r = 0; r < g->n && !rc; ++r) {
        HIP_TRY(hipSetDevice(g->dev[r]));
        rc = spmv_dev_fill_synth(g->x[r], g->N, seed, 0, g->stream[r]);
    }
fail:
    return rc;
}

/*
 * Measured kernel choice for the shards (spmv_*_autotune on every device;
 * device 0's pick is used by all so that the ranks stay in step; a device
 * that did not keep a blocked copy builds one when the pick needs it).
 */
int spmv_mgpu_autotune(spmv_mgpu *g, int *kernel) {
    MG_OK(g);
    if (!g || !kernel)
        return -EINVAL;
    int rc = 0, pick = -1;
    device_guard keep;
    const int blocked = g->is_hll ? SPMV_HLL_KERNEL_PANELS : SPMV_CSR_KERNEL_PANELS;
    if (!g->hll[0] && !g->csr[0])
        return -EINVAL; /* nothing loaded */
    {
        /* one host thread per device: the selectors run side by side (a
         * handle per device, a device per thread -- the concurrency the engine
         * supports, tests/test_gpu_threads.py); one after the other they were
         * ngpus x the 0.5..3 s of a selector run */
        std::vector<int> rcs((size_t)g->n, 0), picks((size_t)g->n, -1);
        std::vector<std::thread> th;
        auto tune = [g, &rcs, &picks](int r) {
            if (hipSetDevice(g->dev[r]) != hipSuccess) {
                rcs[(size_t)r] = -ENODEV;
                return;
            }
            double *yfrag = g->y[r] + (size_t)g->start[r];
            int k = -1;
            rcs[(size_t)r] =
                g->is_hll
                    ? spmv_hll_autotune(g->hll[r], g->x[r], yfrag, 1, &k, NULL)
                    : spmv_csr_autotune(g->csr[r], g->x[r], yfrag, 1, &k, NULL);
            picks[(size_t)r] = k;
        };
        for (int r = 0; r < g->n; ++r) {
            if (!g->hll[r] && !g->csr[r])
                continue; /* an empty range */
            try {
                th.emplace_back(tune, r);
            } catch (...) { /* no thread to be had: tune it from here */
                tune(r);
            }
        }
        for (std::thread &t : th)
            t.join();
        for (int r = 0; r < g->n && !rc; ++r)
            rc = rcs[(size_t)r];
        pick = picks[0];
    }
    /* device 0's blocked copy is the model: every other shard is rebuilt
     * with its schedule, tile height and build options unless it already
     * holds the same (shards of one matrix run one arrangement).  With logical
     * shards a sweep copy is first rebuilt on a grid that leaves reserve_cus
     * compute units to RCCL (the all-gather of shard c runs beside the
     * persistent launch of shard c + 1). */
    if (!rc && pick == blocked && g->L > 1 && g->reserve_cus > 0) {
        spmv_panel_opts o;
        int waves = 0;
        spmv_panel_opts_default(&o); /* struct_size: the layout call checks it */
        HIP_TRY(hipSetDevice(g->dev[0]));
        rc = g->is_hll ? spmv_hll_panels_layout(g->hll[0], &o, &waves)
                       : spmv_csr_panels_layout(g->csr[0], &o, &waves);
        if (!rc && o.sched == 1) {
            o.reserve_cus = g->reserve_cus;
            rc = g->is_hll ? spmv_hll_build_panels_opts(g->hll[0], &o)
                           : spmv_csr_build_panels_opts(g->csr[0], &o);
        }
    }
    for (int r = 0; r < g->n && !rc && pick == blocked; ++r) {
        HIP_TRY(hipSetDevice(g->dev[r]));
        for (int c = 0; c < g->L && !rc; ++c) {
            if ((r == 0 && c == 0) || (!hll_at(g, r, c) && !csr_at(g, r, c)))
                continue;
            bool same = false;
            if (c == 0 && g->L == 1)
                same = g->is_hll
                           ? (spmv_hll_panels_schedule(g->hll[r]) ==
                                  spmv_hll_panels_schedule(g->hll[0]) &&
                              spmv_hll_panels_tile_rows(g->hll[r]) ==
                                  spmv_hll_panels_tile_rows(g->hll[0]))
                           : (spmv_csr_panels_schedule(g->csr[r]) ==
                                  spmv_csr_panels_schedule(g->csr[0]) &&
                              spmv_csr_panels_tile_rows(g->csr[r]) ==
                                  spmv_csr_panels_tile_rows(g->csr[0]));
            if (!same)
                rc = g->is_hll
                         ? spmv_hll_build_panels_like(hll_at(g, r, c), g->hll[0])
                         : spmv_csr_build_panels_like(csr_at(g, r, c), g->csr[0]);
        }
    }
    if (!rc)
        *kernel = pick;
fail:
    return rc;
}

/* the 2-D blocked copy on every shard with the same options (NULL: the
 * defaults): what a caller that names the blocked kernel id needs before
 * spmv_mgpu_spmv / _run (spmv_mgpu_autotune builds it by itself) */
int spmv_mgpu_build_panels(spmv_mgpu *g, const spmv_panel_opts *opts) {
    MG_OK(g);
    int rc = 0;
    device_guard keep;
    for (int r = 0; r < g->n && !rc; ++r) {
        HIP_TRY(hipSetDevice(g->dev[r]));
        for (int c = 0; c < g->L && !rc; ++c) {
            if (!hll_at(g, r, c) && !csr_at(g, r, c))
                continue;
            rc = g->is_hll ? spmv_hll_build_panels_opts(hll_at(g, r, c), opts)
                           : spmv_csr_build_panels_opts(csr_at(g, r, c), opts);
        }
    }
fail:
    return rc;
}

static int launch_shard(spmv_mgpu *g, int r, int kernel);
static int sync_all(spmv_mgpu *g);
static int push_rows(spmv_mgpu *g, int r, int a, int b, int slot);
static int gather_y(spmv_mgpu *g, bool after_kernels = true);
static bool staged(const spmv_mgpu *g, int kernel);
static int step_staged(spmv_mgpu *g, int kernel, hipEvent_t *kernels_done);

static double wall_ms_now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec * 1e3 + (double)t.tv_nsec * 1e-6;
}

/*
 * `iters` steps of (local kernels + all-gather of y); ms_each[i] = wall time
 * of step i with every device synchronised on both sides (max over devices
 * by construction).  kernel < 0: 2 (CSR sub-wave) or 1 (HLL col-major).
 */
int spmv_mgpu_spmv(spmv_mgpu *g, int kernel, int warmup, int iters,
                   double *ms_each) {
    MG_OK(g);
    if (!g || iters < 0 || warmup < 0 || (iters && !ms_each))
        return -EINVAL;
    if (!mg_loaded(g))
        return -EINVAL; /* nothing loaded (or the last load failed) */
    int rc = 0;
    device_guard keep;
    if (kernel < 0)
        kernel = g->is_hll ? 1 : 2;
    for (int it = -warmup; it < iters && !rc; ++it) {
        for (int r = 0; r < g->n; ++r) {
            HIP_TRY(hipSetDevice(g->dev[r]));
            HIP_TRY(hipStreamSynchronize(g->stream[r]));
        }
        const double t0 = wall_ms_now();
        if (staged(g, kernel)) {
            rc = step_staged(g, kernel, NULL);
        } else {
            for (int r = 0; r < g->n && !rc; ++r) {
                HIP_TRY(hipSetDevice(g->dev[r]));
                rc = launch_shard(g, r, kernel);
            }
            if (!rc)
                rc = gather_y(g);
        }
        if (rc)
            break;
        for (int r = 0; r < g->n; ++r) {
            HIP_TRY(hipSetDevice(g->dev[r]));
            HIP_TRY(hipStreamSynchronize(g->stream[r]));
        }
        if (it >= 0)
            ms_each[it] = wall_ms_now() - t0;
    }
fail:
    return rc;
}

/* stage[c][src][i] -> y[src * rows + c * ch + i] */
__global__ void k_unstage(int world, int k, int ch, const double *stage,
                          double *y) {
    const size_t n = (size_t)world * k * ch;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
         t += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(t % ch);
        const size_t q = t / ch;
        const int src = (int)(q % world), c = (int)(q / world);
        y[((size_t)src * k + c) * ch + i] = stage[t];
    }
}

/* does this launch use the staged pipeline?  direct kernels only (the blocked
 * path runs whole shards), rows divisible into chunks of whole hack blocks */
static bool staged(const spmv_mgpu *g, int kernel) {
    const int blocked = g->is_hll ? SPMV_HLL_KERNEL_PANELS : SPMV_CSR_KERNEL_PANELS;
    if (g->engine == 1) /* the copy engine needs no staging: rows in place */
        return false;
    /* logical shards: the shard is the chunk, whatever the kernel */
    if (g->L > 1)
        return g->even && !g->ragged && !g->loopback &&
               (g->n > 1 || g->force_exchange) && g->stage[0] != NULL;
    /* the CSR stream kernel (4) owns a row-block table of the whole shard: a
     * row sub-range would fall back to the sub-wave kernel (spmv_engine.h),
     * which a long row makes 100x slower -- whole-shard launches instead */
    if (!g->is_hll && kernel == 4)
        return false;
    return g->chunks > 1 && g->even && !g->ragged && !g->loopback &&
           kernel != blocked &&
           (g->n > 1 || g->force_exchange) &&
           g->rows_per_gpu % (g->chunks * HACK_SIZE) == 0 && g->stage[0] != NULL;
}

/* one step with the overlapped exchange: every device's chunk-c kernel, then
 * (second stream, after that kernel) the grouped all-gather of chunk c, while
 * the compute streams go on with chunk c+1; at the end the compute streams
 * wait for the last gathers and un-stage.  kernels_done[r] (may be NULL) is
 * recorded on device r's compute stream right after its LAST chunk kernel,
 * i.e. before the wait for the gathers: the kernel time of a device */
static int step_staged(spmv_mgpu *g, int kernel, hipEvent_t *kernels_done) {
    int rc = 0;
    const int k = g->L > 1 ? g->L : g->chunks, ch = g->rows_per_gpu / k;
    for (int c = 0; c < k && !rc; ++c) {
        for (int r = 0; r < g->n && !rc; ++r) {
            HIP_TRY(hipSetDevice(g->dev[r]));
            /* chunk c of device r goes to stage[c][r][0 .. ch): the launch
             * indexes y from local row 0, so hand it (slot - first row) */
            double *slot = g->stage[r] + ((size_t)c * g->n + r) * ch;
            double *ybase = slot - (size_t)c * ch;
            if (g->L > 1) /* a logical shard is a whole matrix: its row 0 */
                rc = g->is_hll
                         ? spmv_hll_launch(hll_at(g, r, c), kernel, NULL,
                                           g->x[r], slot, g->stream[r])
                         : spmv_csr_launch(csr_at(g, r, c), kernel, NULL,
                                           g->x[r], slot, g->stream[r]);
            else
            rc = g->is_hll
                     ? spmv_hll_launch_blocks(g->hll[r], kernel, NULL, g->x[r],
                                              ybase, c * ch / HACK_SIZE,
                                              (c + 1) * ch / HACK_SIZE,
                                              g->stream[r])
                     : spmv_csr_launch_rows(g->csr[r], kernel, NULL, g->x[r],
                                            ybase, c * ch, (c + 1) * ch,
                                            g->stream[r]);
            if (rc)
                break;
            hipEvent_t e = g->ev_k[(size_t)r * MG_MAX_CHUNKS + c];
            HIP_TRY(hipEventRecord(e, g->stream[r]));
            HIP_TRY(hipStreamWaitEvent(g->xstream[r], e, 0));
            if (c == k - 1 && kernels_done)
                HIP_TRY(hipEventRecord(kernels_done[r], g->stream[r]));
        }
        if (rc)
            break;
        NCCL_TRY(ncclGroupStart());
        {
            ncclResult_t first_bad = ncclSuccess;
            for (int r = 0; r < g->n; ++r) {
                double *blk = g->stage[r] + (size_t)c * g->n * ch;
                const ncclResult_t e = ncclAllGather(
                    blk + (size_t)r * ch, blk, (size_t)ch, ncclDouble,
                    g->comm[r], g->xstream[r]);
                if (e != ncclSuccess && first_bad == ncclSuccess)
                    first_bad = e;
            }
            const ncclResult_t closed = ncclGroupEnd();
            NCCL_TRY(first_bad);
            NCCL_TRY(closed);
        }
    }
    for (int r = 0; r < g->n && !rc; ++r) {
        HIP_TRY(hipSetDevice(g->dev[r]));
        hipEvent_t e = g->ev_x[(size_t)r * MG_MAX_CHUNKS];
        HIP_TRY(hipEventRecord(e, g->xstream[r]));
        HIP_TRY(hipStreamWaitEvent(g->stream[r], e, 0));
        hipLaunchKernelGGL(k_unstage, dim3(2048), dim3(256), 0, g->stream[r],
                           g->n, k, ch, g->stage[r], g->y[r]);
        HIP_TRY(hipGetLastError());
    }
fail:
    return rc;
}

static int launch_shard(spmv_mgpu *g, int r, int kernel) {
    double *yfrag = g->y[r] + (size_t)g->start[r];
    if (!g->hll[r] && !g->csr[r])
        return 0; /* an empty range */
    if (g->L > 1) { /* logical shards, one after the other, in row order */
        const size_t per = (size_t)g->rows_per_gpu / g->L;
        const bool push = g->engine == 1 && (g->n > 1 || g->force_exchange);
        for (int c = 0; c < g->L; ++c) {
            int rc =
                g->is_hll ? spmv_hll_launch(hll_at(g, r, c), kernel, NULL,
                                            g->x[r], yfrag + c * per,
                                            g->stream[r])
                          : spmv_csr_launch(csr_at(g, r, c), kernel, NULL,
                                            g->x[r], yfrag + c * per,
                                            g->stream[r]);
            /* copy engine: shard c travels while shard c + 1 computes */
            if (!rc && push)
                rc = push_rows(g, r, g->start[r] + (int)(c * per),
                               g->start[r] + (int)((c + 1) * per), c);
            if (rc)
                return rc;
        }
        return 0;
    }
    return g->is_hll ? spmv_hll_launch(g->hll[r], kernel, NULL, g->x[r], yfrag,
                                       g->stream[r])
                     : spmv_csr_launch(g->csr[r], kernel, NULL, g->x[r], yfrag,
                                       g->stream[r]);
}

/*
 * Ragged fragments (nnz-balanced partition).  Every variant is ONE RCCL group
 * over all devices; every call inside the group is checked and the group is
 * always closed before an error leaves.
 *   P2P     device r sends its fragment to each peer and receives each peer's
 *           fragment straight into place (ncclSend / ncclRecv): exactly M - own
 *           rows arrive per device, each xGMI link carries one peer's fragment
 *   BCAST   one in-place ncclBroadcast per device (SURVEY 8e "general case")
 *   PADDED  the fragment is copied to slot r of a [n][longest] staging buffer,
 *           one in-place ncclAllGather of `longest` rows, then the peers'
 *           fragments are copied back into row order
 */
static int gather_y_ragged(spmv_mgpu *g) {
    int rc = 0;
    ncclResult_t first_bad = ncclSuccess;
#define GROUPED(call)                                                         \
    do {                                                                      \
        const ncclResult_t e_ = (call);                                       \
        if (e_ != ncclSuccess && first_bad == ncclSuccess)                    \
            first_bad = e_;                                                   \
    } while (0)
    const int L = g->longest;
    if (g->xchg == SPMV_MGPU_XCHG_PADDED) {
        if (!g->pad[0])
            return -EINVAL;
        for (int r = 0; r < g->n; ++r) {
            const size_t cnt = (size_t)(g->start[r + 1] - g->start[r]);
            HIP_TRY(hipSetDevice(g->dev[r]));
            if (cnt)
                HIP_TRY(hipMemcpyAsync(g->pad[r] + (size_t)r * L,
                                       g->y[r] + g->start[r],
                                       cnt * sizeof(double),
                                       hipMemcpyDeviceToDevice, g->stream[r]));
        }
    }
    NCCL_TRY(ncclGroupStart());
    for (int r = 0; r < g->n; ++r) {
        if (g->xchg == SPMV_MGPU_XCHG_PADDED) {
            GROUPED(ncclAllGather(g->pad[r] + (size_t)r * L, g->pad[r],
                                  (size_t)L, ncclDouble, g->comm[r],
                                  g->stream[r]));
            continue;
        }
        for (int p = 0; p < g->n; ++p) {
            const size_t cnt = (size_t)(g->start[p + 1] - g->start[p]);
            double *frag = g->y[r] + g->start[p];
            if (!cnt)
                continue;
            if (g->xchg == SPMV_MGPU_XCHG_BCAST) {
                GROUPED(ncclBroadcast(frag, frag, cnt, ncclDouble, p,
                                      g->comm[r], g->stream[r]));
            } else if (p == r) { /* P2P: my fragment to every peer */
                for (int q = 0; q < g->n; ++q)
                    if (q != r)
                        GROUPED(ncclSend(frag, cnt, ncclDouble, q, g->comm[r],
                                         g->stream[r]));
            } else {
                GROUPED(ncclRecv(frag, cnt, ncclDouble, p, g->comm[r],
                                 g->stream[r]));
            }
        }
    }
    {
        const ncclResult_t closed = ncclGroupEnd();
        NCCL_TRY(first_bad);
        NCCL_TRY(closed);
    }
#undef GROUPED
    if (g->xchg == SPMV_MGPU_XCHG_PADDED) {
        for (int r = 0; r < g->n; ++r) {
            HIP_TRY(hipSetDevice(g->dev[r]));
            for (int p = 0; p < g->n; ++p) {
                const size_t cnt = (size_t)(g->start[p + 1] - g->start[p]);
                if (p != r && cnt)
                    HIP_TRY(hipMemcpyAsync(
                        g->y[r] + g->start[p], g->pad[r] + (size_t)p * L,
                        cnt * sizeof(double), hipMemcpyDeviceToDevice,
                        g->stream[r]));
            }
        }
    }
fail:
    return rc;
}

/* COPY ENGINE.  push_rows: after what device r has enqueued on its compute
 * stream so far (the kernel that produced rows [a, b), global numbering), its
 * second stream copies those rows into every peer's y.  `slot` picks the event
 * (one per logical shard of a step). */
static int push_rows(spmv_mgpu *g, int r, int a, int b, int slot) {
    int rc = 0;
    if (b <= a)
        return 0;
    hipEvent_t e = g->ev_k[(size_t)r * MG_MAX_CHUNKS + slot];
    HIP_TRY(hipSetDevice(g->dev[r]));
    HIP_TRY(hipEventRecord(e, g->stream[r]));
    HIP_TRY(hipStreamWaitEvent(g->xstream[r], e, 0));
    for (int p = 0; p < g->n; ++p)
        if (p != r)
            HIP_TRY(hipMemcpyAsync(g->y[p] + a, g->y[r] + a,
                                   (size_t)(b - a) * sizeof(double),
                                   hipMemcpyDefault, g->xstream[r]));
fail:
    return rc;
}

/* ... and the close of a step: every device's compute stream waits for its own
 * pushes (the next step's kernel overwrites the rows they read; synchronising
 * the compute streams of ALL devices then covers every copy of the step) */
static int close_pushes(spmv_mgpu *g) {
    int rc = 0;
    for (int r = 0; r < g->n; ++r) {
        hipEvent_t e = g->ev_x[(size_t)r * MG_MAX_CHUNKS];
        HIP_TRY(hipSetDevice(g->dev[r]));
        HIP_TRY(hipEventRecord(e, g->xstream[r]));
        HIP_TRY(hipStreamWaitEvent(g->stream[r], e, 0));
    }
fail:
    return rc;
}

static int gather_y_copy(spmv_mgpu *g, bool after_kernels) {
    int rc = 0;
    /* with logical shards the pushes were issued shard by shard behind the
     * kernels (launch_shard); otherwise -- and when the exchange runs by
     * itself -- the whole fragment goes now */
    for (int r = 0; r < g->n && !rc && (g->L == 1 || !after_kernels); ++r)
        rc = push_rows(g, r, g->start[r], g->start[r + 1], 0);
    if (!rc)
        rc = close_pushes(g);
    return rc;
}

/* one grouped in-place all-gather of y over all devices (n > 1) */
static int gather_y(spmv_mgpu *g, bool after_kernels) {
    int rc = 0;
    if (g->n < 2 && !g->force_exchange)
        return 0;
    if (g->engine == 1)
        return gather_y_copy(g, after_kernels);
    if (g->ragged)
        return gather_y_ragged(g);
    /* every call inside the group is checked, and the group is ALWAYS closed
     * before an error leaves this function */
    NCCL_TRY(ncclGroupStart());
    {
        ncclResult_t first_bad = ncclSuccess;
        for (int r = 0; r < g->n; ++r) {
            const ncclResult_t e = ncclAllGather(
                g->y[r] + (size_t)r * g->rows_per_gpu, g->y[r],
                (size_t)g->rows_per_gpu, ncclDouble, g->comm[r], g->stream[r]);
            if (e != ncclSuccess && first_bad == ncclSuccess)
                first_bad = e;
        }
        const ncclResult_t closed = ncclGroupEnd();
        NCCL_TRY(first_bad);
        NCCL_TRY(closed);
    }
fail:
    return rc;
}

static int sync_all(spmv_mgpu *g) {
    int rc = 0;
    for (int r = 0; r < g->n; ++r) {
        HIP_TRY(hipSetDevice(g->dev[r]));
        HIP_TRY(hipStreamSynchronize(g->stream[r]));
        HIP_TRY(hipStreamSynchronize(g->xstream[r]));
    }
fail:
    return rc;
}

/* chunks: 1 = the all-gather follows the kernels (in place); 2..16 = staged,
 * overlapped (direct kernels only; rows per device must split into chunks of
 * whole hack blocks, else the launch falls back to 1).  force != 0: run the
 * collective even with ONE device (a 1-rank all-gather: lets the staging and
 * un-staging logic run on a 1-GPU box). */
int spmv_mgpu_set_exchange(spmv_mgpu *g, int chunks, int force) {
    MG_OK(g);
    if (chunks < 1 || chunks > MG_MAX_CHUNKS)
        return -EINVAL;
    int rc = 0;
    device_guard keep;
    g->chunks = chunks;
    g->force_exchange = force != 0;
    const size_t ny = g->ragged ? 0 : y_len(g);
    for (int r = 0; r < g->n; ++r) {
        HIP_TRY(hipSetDevice(g->dev[r]));
        (void)hipFree(g->stage[r]);
        g->stage[r] = NULL;
        if ((chunks > 1 || g->L > 1) && ny > 0)
            HIP_TRY(hipMalloc((void **)&g->stage[r], ny * sizeof(double)));
    }
fail:
    return rc;
}

/* which engine moves the fragments: SPMV_MGPU_ENGINE_RCCL (collectives) or
 * _COPY (peer copies on the copy engines; see struct spmv_mgpu) */
int spmv_mgpu_set_exchange_engine(spmv_mgpu *g, int engine) {
    MG_OK(g);
    if (engine != SPMV_MGPU_ENGINE_RCCL && engine != SPMV_MGPU_ENGINE_COPY)
        return -EINVAL;
    if (g->loopback && engine != SPMV_MGPU_ENGINE_COPY)
        return -ENOTSUP; /* a rehearsal handle has no communicator */
    device_guard keep;
    const int rc = sync_all(g);
    if (!rc)
        g->engine = engine;
    return rc;
}

/* L logical shards per device from the NEXT load / generate on (1 = off);
 * reserve_cus: compute units a sweep copy leaves to RCCL when L > 1 (applied
 * by spmv_mgpu_autotune).  The load answers -EINVAL when the rows of a device
 * do not split into L shards of whole hack blocks (or the partition is not the
 * even one). */
int spmv_mgpu_set_logical_shards(spmv_mgpu *g, int shards, int reserve_cus) {
    MG_OK(g);
    if (shards < 1 || shards > MG_MAX_CHUNKS || reserve_cus < 0)
        return -EINVAL;
    g->want_L = shards;
    g->reserve_cus = reserve_cus;
    return 0;
}

/* how RAGGED fragments travel (see gather_y_ragged); the setting persists
 * over reloads; an even partition keeps its in-place all-gather */
int spmv_mgpu_set_ragged_exchange(spmv_mgpu *g, int kind) {
    MG_OK(g);
    if (kind != SPMV_MGPU_XCHG_P2P && kind != SPMV_MGPU_XCHG_BCAST &&
        kind != SPMV_MGPU_XCHG_PADDED)
        return -EINVAL;
    device_guard keep;
    const int rc0 = sync_all(g);
    if (rc0)
        return rc0;
    g->xchg = kind;
    return alloc_pad(g);
}

/* the partition in use: starts[ngpus + 1] row offsets, entries[ngpus] true
 * entries per device (either may be NULL); returns 1 when the ranges differ
 * (ragged fragments), 0 for the even partition */
int spmv_mgpu_partition(const spmv_mgpu *g, int *starts, int64_t *entries) {
    MG_OK(g);
    for (int r = 0; r <= g->n && starts; ++r)
        starts[r] = g->start[r];
    for (int r = 0; r < g->n && entries; ++r) {
        int64_t nz = 0;
        spmv_mgpu *m = const_cast<spmv_mgpu *>(g);
        for (int c = 0; c < g->L; ++c)
            nz += hll_at(m, r, c) ? hll_at(m, r, c)->NZ
                                  : (csr_at(m, r, c) ? csr_at(m, r, c)->NZ : 0);
        entries[r] = nz;
    }
    return g->ragged;
}

int spmv_mgpu_run(spmv_mgpu *g, int kernel, int warmup, int steps,
                  double *wall_ms_total, double *kernel_ms_avg) {
    MG_OK(g);
    if (!g || steps < 1 || warmup < 0 || !wall_ms_total)
        return -EINVAL;
    if (!mg_loaded(g))
        return -EINVAL;
    int rc = 0;
    device_guard keep;
    if (kernel < 0)
        kernel = g->is_hll ? 1 : 2;
    /* one event pair per device and step */
    std::vector<hipEvent_t> ev((size_t)g->n * steps * 2, NULL);
    for (int r = 0; r < g->n; ++r) {
        HIP_TRY(hipSetDevice(g->dev[r]));
        for (int k = 0; k < 2 * steps; ++k)
            HIP_TRY(hipEventCreate(&ev[((size_t)r * steps) * 2 + k]));
    }
    for (int it = -warmup; it < steps && !rc; ++it) {
        if (it == 0) {
            rc = sync_all(g);
            if (rc)
                break;
            *wall_ms_total = wall_ms_now();
        }
        if (staged(g, kernel)) {
            /* events around a device's chunk kernels: from before the first
             * to right after the last, BEFORE its stream waits for the gathers
             * (those overlap the kernels; the un-staging copy is exchange) */
            std::vector<hipEvent_t> done((size_t)g->n, NULL);
            for (int r = 0; r < g->n && it >= 0; ++r) {
                HIP_TRY(hipSetDevice(g->dev[r]));
                HIP_TRY(hipEventRecord(ev[((size_t)r * steps + it) * 2],
                                       g->stream[r]));
                done[r] = ev[((size_t)r * steps + it) * 2 + 1];
            }
            rc = step_staged(g, kernel, it >= 0 ? done.data() : NULL);
            continue;
        }
        for (int r = 0; r < g->n && !rc; ++r) {
            HIP_TRY(hipSetDevice(g->dev[r]));
            if (it >= 0)
                HIP_TRY(hipEventRecord(ev[((size_t)r * steps + it) * 2],
                                       g->stream[r]));
            rc = launch_shard(g, r, kernel);
            if (!rc && it >= 0)
                HIP_TRY(hipEventRecord(ev[((size_t)r * steps + it) * 2 + 1],
                                       g->stream[r]));
        }
        if (!rc)
            rc = gather_y(g);
    }
    if (!rc)
        rc = sync_all(g);
    if (!rc) {
        *wall_ms_total = wall_ms_now() - *wall_ms_total;
        for (int r = 0; r < g->n && kernel_ms_avg; ++r) {
            double acc = 0.0;
            for (int it = 0; it < steps; ++it) {
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(
                    &ms, ev[((size_t)r * steps + it) * 2],
                    ev[((size_t)r * steps + it) * 2 + 1]));
                acc += ms;
            }
            kernel_ms_avg[r] = acc / steps;
        }
    }
fail:
    for (size_t k = 0; k < ev.size(); ++k)
        if (ev[k])
            (void)hipEventDestroy(ev[k]);
    return rc;
}

int spmv_mgpu_exchange_only(spmv_mgpu *g, int iters, double *ms_avg) {
    MG_OK(g);
    if (!g || iters < 1 || !ms_avg)
        return -EINVAL;
    if (!mg_loaded(g))
        return -EINVAL;
    device_guard keep;
    *ms_avg = 0.0;
    if (g->n < 2 && !g->force_exchange)
        return 0;
    int rc = gather_y(g, false); /* warm */
    if (!rc)
        rc = sync_all(g);
    const double t0 = wall_ms_now();
    for (int it = 0; it < iters && !rc; ++it)
        rc = gather_y(g, false);
    if (!rc)
        rc = sync_all(g);
    if (!rc)
        *ms_avg = (wall_ms_now() - t0) / iters;
    return rc;
}

int spmv_mgpu_rccl_version(void) {
    int v = 0;
    return ncclGetVersion(&v) == ncclSuccess ? v : -EIO;
}

int spmv_mgpu_comm_ranks(const spmv_mgpu *g) {
    MG_OK(g);
    if (g->loopback)
        return 0; /* a rehearsal handle has no communicator */
    if (!g || g->comm.empty() || !g->comm[0])
        return -EINVAL;
    int n = 0;
    return ncclCommCount(g->comm[0], &n) == ncclSuccess ? n : -EIO;
}

int spmv_mgpu_device_bus_id(const spmv_mgpu *g, int rank, char *buf,
                            size_t len) {
    MG_OK(g);
    if (!g || rank < 0 || rank >= g->n)
        return -EINVAL;
    return spmv_device_pci_bus_id(g->dev[rank], buf, len);
}

int spmv_mgpu_shard_info(const spmv_mgpu *g, int rank, int64_t *stored,
                         int64_t *alg_bytes, char *layout, size_t len) {
    MG_OK(g);
    if (!g || rank < 0 || rank >= g->n || (!g->hll[rank] && !g->csr[rank]))
        return -EINVAL;
    {
        /* summed over the device's logical shards */
        spmv_mgpu *m = const_cast<spmv_mgpu *>(g);
        int64_t st = 0, by = 0;
        for (int c = 0; c < g->L; ++c) {
            if (hll_at(m, rank, c)) {
                st += hll_at(m, rank, c)->slots;
                by += spmv_hll_algorithmic_bytes(hll_at(m, rank, c));
            } else if (csr_at(m, rank, c)) {
                st += csr_at(m, rank, c)->NZ;
                by += spmv_csr_algorithmic_bytes(csr_at(m, rank, c));
            }
        }
        if (stored)
            *stored = st;
        if (alg_bytes)
            *alg_bytes = by;
    }
    if (layout && len) {
        layout[0] = 0;
        const int rc = g->hll[rank]
                           ? spmv_hll_panels_describe(g->hll[rank], layout, len)
                           : spmv_csr_panels_describe(g->csr[rank], layout, len);
        if (rc && rc != -ENOENT)
            return rc;
    }
    return 0;
}

/* the gathered y as device `rank` holds it (M doubles) */
int spmv_mgpu_get_y(spmv_mgpu *g, int rank, double *y_host) {
    MG_OK(g);
    if (!g || rank < 0 || rank >= g->n || !y_host)
        return -EINVAL;
    if (!mg_loaded(g) || !g->y[rank])
        return -EINVAL;
    device_guard keep;
    HIP_RET(hipSetDevice(g->dev[rank]));
    HIP_RET(hipMemcpy(y_host, g->y[rank], (size_t)g->M * sizeof(double),
                      hipMemcpyDeviceToHost));
    return 0;
}

int spmv_mgpu_info(const spmv_mgpu *g, int *ngpus, int *rows_per_gpu,
                   int64_t *nnz_total, int64_t *bytes_per_gpu) {
    MG_OK(g);
    if (!g)
        return -EINVAL;
    int64_t nz = 0, by = 0;
    spmv_mgpu *m = const_cast<spmv_mgpu *>(g);
    for (int r = 0; r < g->n; ++r) {
        int64_t b = 0;
        for (int c = 0; c < g->L; ++c) {
            if (hll_at(m, r, c)) {
                nz += hll_at(m, r, c)->NZ;
                b += spmv_hll_algorithmic_bytes(hll_at(m, r, c));
            } else if (csr_at(m, r, c)) {
                nz += csr_at(m, r, c)->NZ;
                b += spmv_csr_algorithmic_bytes(csr_at(m, r, c));
            }
        }
        if (b > by) /* the heaviest device bounds the step */
            by = b;
    }
    if (ngpus)
        *ngpus = g->n;
    if (rows_per_gpu)
        *rows_per_gpu = g->rows_per_gpu;
    if (nnz_total)
        *nnz_total = nz;
    if (bytes_per_gpu)
        *bytes_per_gpu = by;
    return 0;
}

} /* extern "C" */

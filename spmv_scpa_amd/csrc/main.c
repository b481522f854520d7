/*
 * main.c -- the spmv_scpa_amd benchmark driver.
 *
 * Same job and flags as the reference driver (src/main.c:28-109, 361-379):
 * load a matrix, build both HLL layouts, draw x, run the serial, OpenMP and
 * GPU benchmark grid, optionally validate everything against serial CSR
 * (-d), and append rows to serial.csv / omp.csv / cuda.csv in -o.  Added:
 * synthetic matrices (-s), warm multi-iteration GPU timing with roofline
 * figures (roofline.csv), and clean errors where the reference crashes
 * (missing file: main.c:79 tests `!A` on an ERR_PTR).
 */
#include <errno.h>
#include <getopt.h>
#include <libgen.h>
#include <math.h>
#include <omp.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "csr.h"
#include "err.h"
#include "hip_csr.h"
#include "hip_hll.h"
#include "hll.h"
#include "logger.h"
#include "spmv_engine.h"
#include "spmv_mgpu.h"
#include "spmv_synth.h"
#include "utils.h"

static struct {
    const char *matrix, *out_dir, *synthetic;
    int rows, nnz_row, iters, cpu, gpus, only_mgpu;
    long long window;
    bool debug;
    int partition; /* SPMV_MGPU_PART_EVEN / _NNZ */
    int xchg;      /* SPMV_MGPU_XCHG_*: how ragged fragments travel */
    int chunks;    /* > 1: opt-in staged exchange (even partition only) */
    int shards;    /* > 1: opt-in logical shards per GPU (overlap for the
                      blocked path too; even partition only) */
    int engine;    /* SPMV_MGPU_ENGINE_RCCL (default) / _COPY */
} opt = {NULL, NULL, NULL, 1000000, 16, 20, 1, 1, 0, 0, false,
         SPMV_MGPU_PART_EVEN, SPMV_MGPU_XCHG_P2P, 1, 1, SPMV_MGPU_ENGINE_RCCL};

static sparse_csr *A;
static sparse_hll *H_row, *H_col;
static vec x, expected;
static double *row_scale; /* sum_j |a_ij x_j| for the strict check */
static bool logger_open;

static void cleanup(void) {
    vec_put(&expected);
    free(row_scale);
    row_scale = NULL;
    if (!IS_ERR_OR_NULL(H_col))
        hll_free(H_col);
    if (!IS_ERR_OR_NULL(H_row))
        hll_free(H_row);
    vec_put(&x);
    if (!IS_ERR_OR_NULL(A))
        csr_free(A);
    if (logger_open)
        logger_close();
}

static void die(const char *what, int code) {
    LOG_ERR("%s (err %d: %s)", what, code, strerror(code < 0 ? -code : code));
    cleanup();
    exit(EXIT_FAILURE);
}

/* -d: the reference's criterion (abs L2 <= 0.1) AND this build's bound
 * (1e-6 relative) */
static void check(const char *label, const vec y) {
    if (!opt.debug)
        return;
    double e = max_rel_err(expected, y, row_scale);
    if (validation_vec_result(expected, y) != 0 || !(e <= 1e-6)) {
        LOG_ERR("[%s] validation failed (max rel err %.3e)", label, e);
        cleanup();
        exit(EXIT_FAILURE);
    }
}

static void run_serial(void) {
    bench r;
    int rc = bench_csr_serial(A, x.data, &r);
    if (rc)
        die("[CSR serial] failed", rc);
    log_csr_serial_benchmark(A, r);
    if (opt.debug) {
        expected = r.data; /* kept for every later comparison */
        row_scale = malloc((size_t)A->M * sizeof *row_scale);
        if (!row_scale)
            die("row scale", -ENOMEM);
        for (int i = 0; i < A->M; ++i) {
            double s = 0.0;
            for (int k = A->IRP[i]; k < A->IRP[i + 1]; ++k)
                s += fabs(A->AS[k] * x.data[A->JA[k]]);
            row_scale[i] = s;
        }
    } else {
        vec_put(&r.data);
    }
    rc = bench_hll_serial(H_row, x.data, &r);
    if (rc)
        die("[HLL serial] failed", rc);
    log_hll_serial_benchmark(H_row, r);
    check("HLL serial", r.data);
    vec_put(&r.data);
}

static void run_omp(void) {
    /* the reference's thread ladder (main.c:177-180) plus all cores; counts
     * above the host's limit are skipped (the reference asserts) */
    int ladder[7] = {2, 4, 8, 16, 32, 40, omp_get_num_procs()};
    int limit = omp_get_max_threads();
    size_t count = ARRAY_SIZE(ladder);
    for (size_t k = 0; k + 1 < ARRAY_SIZE(ladder); ++k)
        if (ladder[k] == ladder[ARRAY_SIZE(ladder) - 1])
            count = ARRAY_SIZE(ladder) - 1; /* all-cores entry is a repeat */
    for (size_t k = 0; k < count; ++k) {
        int t = ladder[k];
        if (t > limit || t < 1)
            continue;
        OMP_WARMUP(t);
        bench_omp r = {.num_threads = t};
        int rc = bench_csr_omp_nnz_balancing(A, x.data, &r);
        if (rc)
            die("[CSR omp_nnz] failed", rc);
        log_csr_omp_benchmark(A, r);
        check("CSR omp_nnz", r.bench.data);
        vec_put(&r.bench.data);

        r.num_threads = t;
        rc = bench_csr_omp_guided(A, x.data, &r);
        if (rc)
            die("[CSR omp_guided] failed", rc);
        log_csr_omp_benchmark(A, r);
        check("CSR omp_guided", r.bench.data);
        vec_put(&r.bench.data);

        r.num_threads = t;
        rc = bench_hll_omp(H_row, x.data, &r);
        if (rc)
            die("[HLL omp] failed", rc);
        log_hll_omp_benchmark(H_row, r);
        check("HLL omp", r.bench.data);
        vec_put(&r.bench.data);
    }
}

typedef int (*csr_hip_fn)(const sparse_csr *, const double *, bench_hip *);
typedef int (*hll_hip_fn)(const sparse_hll *, const double *, bench_hip *);

static int cmp_double(const void *a, const void *b) {
    double u = *(const double *)a, v = *(const double *)b;
    return (u > v) - (u < v);
}

static double median_of(double *v, int n) {
    qsort(v, (size_t)n, sizeof *v, cmp_double);
    return (n & 1) ? v[n / 2] : 0.5 * (v[n / 2 - 1] + v[n / 2]);
}

static void run_gpu(void) {
    static const csr_hip_fn csr_k[SPMV_NUM_CSR_KERNELS] = {
        bench_csr_hip_thread_row, bench_csr_hip_wave_row,
        bench_csr_hip_subwave_row, bench_csr_hip_block_row,
        bench_csr_hip_stream};
    static const hll_hip_fn hll_k[SPMV_NUM_HLL_KERNELS] = {
        bench_hll_hip_threads_row_major, bench_hll_hip_threads_col_major,
        bench_hll_hip_wave_block, bench_hll_hip_subwave_row};
    static const int wpb[] = {2, 4, 8}; /* reference main.c:265-269 */
    char label[64];

    /* one-shot grid, the reference's protocol: cuda.csv */
    for (int kid = 0; kid < SPMV_NUM_CSR_KERNELS; ++kid)
        for (size_t w = 0; w < ARRAY_SIZE(wpb); ++w) {
            bench_hip r = {.waves_per_block = wpb[w]};
            int rc = csr_k[kid](A, x.data, &r);
            snprintf(label, sizeof label, "CSR HIP kernel %d wpb %d", kid,
                     wpb[w]);
            if (rc)
                die(label, rc);
            check(label, r.bench.data);
            vec_put(&r.bench.data);
            log_csr_hip_benchmark(A, r, kid);
        }
    for (int kid = 0; kid < SPMV_NUM_HLL_KERNELS; ++kid) {
        const sparse_hll *H = (kid == 0 || kid == 3) ? H_row : H_col;
        for (size_t w = 0; w < ARRAY_SIZE(wpb); ++w) {
            bench_hip r = {.waves_per_block = wpb[w]};
            int rc = hll_k[kid](H, x.data, &r);
            snprintf(label, sizeof label, "HLL HIP kernel %d wpb %d", kid,
                     wpb[w]);
            if (rc)
                die(label, rc);
            check(label, r.bench.data);
            vec_put(&r.bench.data);
            log_hll_hip_benchmark(H, r, kid);
        }
    }

    /* warm, resident timing: upload once, median of opt.iters launches */
    if (opt.iters <= 0)
        return;
    double *ms = malloc((size_t)opt.iters * sizeof *ms);
    double *d_x = NULL, *d_y = NULL;
    spmv_csr_dev *dA = NULL;
    spmv_hll_dev *dHr = NULL, *dHc = NULL;
    int rc = ms ? 0 : -ENOMEM;
    if (!rc)
        rc = spmv_dev_malloc((void **)&d_x, (size_t)A->N * sizeof(double));
    if (!rc)
        rc = spmv_dev_malloc((void **)&d_y, (size_t)A->M * sizeof(double));
    if (!rc)
        rc = spmv_copy_h2d(d_x, x.data, (size_t)A->N * sizeof(double));
    if (!rc)
        rc = spmv_csr_upload(A, &dA);
    if (!rc)
        rc = spmv_hll_upload(H_row, 0, &dHr);
    if (!rc)
        rc = spmv_hll_upload(H_col, 1, &dHc);
    /* working sets under 512 MB are flushed out of the 256 MiB Infinity
     * Cache by a read-only sweep of 1 GiB between timed launches */
    size_t flush = spmv_csr_algorithmic_bytes(dA) < (512ll << 20)
                       ? (size_t)1 << 30 : 0;
    for (int kid = 0; !rc && kid < SPMV_NUM_CSR_KERNELS; ++kid) {
        spmv_launch_opts o = {.waves_per_block = 4};
        rc = spmv_csr_time(dA, kid, &o, d_x, d_y, 3, opt.iters, flush, ms, NULL);
        if (!rc)
            log_roofline(A->name, "CSR", kid, 4, 1, A->M, A->N, A->NZ, A->NZ,
                         spmv_csr_algorithmic_bytes(dA),
                         median_of(ms, opt.iters));
    }
    for (int kid = 0; !rc && kid < SPMV_NUM_HLL_KERNELS; ++kid) {
        spmv_hll_dev *dH = (kid == 0 || kid == 3) ? dHr : dHc;
        spmv_launch_opts o = {.waves_per_block = 4};
        rc = spmv_hll_time(dH, kid, &o, d_x, d_y, 3, opt.iters, flush, ms, NULL);
        if (!rc)
            log_roofline(A->name, "HLL", kid, 4, 1, A->M, A->N, A->NZ,
                         hll_num_slots(H_row), spmv_hll_algorithmic_bytes(dH),
                         median_of(ms, opt.iters));
    }
    /* measured kernel choice; the 2-D blocked path is built only if the
     * coalesced kernels run far below the stream rate */
    if (!rc) {
        int best = -1;
        double best_ms = 0.0;
        rc = spmv_csr_autotune(dA, d_x, d_y, 1, &best, &best_ms);
        if (!rc) {
            LOG_INFO("CSR autotune: kernel %d, %.4f ms, %.1f GFLOP/s", best,
                     best_ms, compute_gflops64(best_ms, A->NZ));
            if (best == SPMV_CSR_KERNEL_PANELS)
                log_roofline(A->name, "CSR", best, 8, 1, A->M, A->N, A->NZ,
                             A->NZ, spmv_csr_algorithmic_bytes(dA), best_ms);
        }
    }
    if (!rc) {
        int best = -1;
        double best_ms = 0.0;
        rc = spmv_hll_autotune(dHc, d_x, d_y, 1, &best, &best_ms);
        if (!rc) {
            LOG_INFO("HLL autotune: kernel %d, %.4f ms, %.1f GFLOP/s", best,
                     best_ms, compute_gflops64(best_ms, A->NZ));
            if (best == SPMV_HLL_KERNEL_PANELS)
                log_roofline(A->name, "HLL", best, 8, 1, A->M, A->N, A->NZ,
                             hll_num_slots(H_col),
                             spmv_hll_kernel_bytes(dHc, best), best_ms);
        }
    }
    spmv_hll_release(dHc);
    spmv_hll_release(dHr);
    spmv_csr_release(dA);
    spmv_dev_free(d_y);
    spmv_dev_free(d_x);
    free(ms);
    if (rc)
        die("resident GPU timing failed", rc);
}

/*
 * -g N: the matrix is cut into N contiguous row ranges (multiples of 32
 * rows), one per GPU -- equal row counts, or with `--partition nnz`
 * near-equal entry counts (the multi-GPU form of the reference's
 * partition_csr_rows, csr.c:218-276) -- a step = every shard's kernel + the
 * exchange of the y fragments over RCCL: one in-place all-gather (even
 * partition) or the ragged exchange chosen by --ragged-exchange.  The
 * overlapped "staged" exchange is opt-in (--exchange-chunks k): it has only
 * ever run as a 1-rank collective (spmv_mgpu.h).  Rows go to roofline.csv
 * with gpus = N.
 */
static void run_multi_gpu(void) {
    spmv_mgpu *g = NULL;
    int rc = spmv_mgpu_create(opt.gpus, &g);
    if (rc)
        die("multi-GPU setup (RCCL communicator)", rc);
    double *ms = malloc((size_t)(opt.iters > 0 ? opt.iters : 1) * sizeof *ms);
    for (int fmt = 0; fmt < 2 && !rc; ++fmt) { /* 0: CSR, 1: HLL */
        if (fmt == 1) { /* fresh communicator and shards per format */
            spmv_mgpu_destroy(g);
            g = NULL;
            rc = spmv_mgpu_create(opt.gpus, &g);
            if (rc)
                break;
        }
        rc = spmv_mgpu_set_ragged_exchange(g, opt.xchg);
        if (!rc)
            rc = spmv_mgpu_set_exchange_engine(g, opt.engine);
        if (!rc) /* a sweep pick leaves 8 CUs to RCCL's kernels; the copy
                    engines need none (spmv_mgpu.h) */
            rc = spmv_mgpu_set_logical_shards(
                g, opt.shards, opt.engine == SPMV_MGPU_ENGINE_RCCL ? 8 : 0);
        if (!rc) {
            rc = spmv_mgpu_load_csr_part(g, A, fmt, opt.partition);
            if (rc == -EINVAL && opt.shards > 1) {
                LOG_WARN("rows per GPU do not split into %d logical shards of "
                         "whole hack blocks: running one shard per GPU",
                         opt.shards);
                rc = spmv_mgpu_set_logical_shards(g, 1, 0);
                if (!rc)
                    rc = spmv_mgpu_load_csr_part(g, A, fmt, opt.partition);
            }
        }
        if (!rc) /* 1: the exchange follows the kernels (default) */
            rc = spmv_mgpu_set_exchange(g, opt.chunks, 0);
        if (!rc)
            rc = spmv_mgpu_set_x(g, x.data);
        if (!rc && fmt == 0) { /* what each device holds */
            int starts[65];
            int64_t ent[64], lo = INT64_MAX, hi = 0;
            const int ragged = spmv_mgpu_partition(g, starts, ent);
            for (int r = 0; r < opt.gpus; ++r) {
                LOG_INFO("GPU %d: rows [%d, %d), %lld entries", r, starts[r],
                         starts[r + 1], (long long)ent[r]);
                if (ent[r] > hi)
                    hi = ent[r];
                if (ent[r] < lo)
                    lo = ent[r];
            }
            LOG_INFO("partition %s%s: entries per GPU max / min = %.3f",
                     opt.partition == SPMV_MGPU_PART_NNZ ? "nnz" : "even",
                     ragged > 0 ? " (ragged fragments)" : "",
                     lo > 0 ? (double)hi / (double)lo : 0.0);
        }
        int kernel = -1;
        if (!rc)
            rc = spmv_mgpu_autotune(g, &kernel);
        if (!rc)
            rc = spmv_mgpu_spmv(g, kernel, 3, opt.iters > 0 ? opt.iters : 1, ms);
        if (rc)
            break;
        if (opt.debug) { /* every device must hold the whole, correct y */
            vec y = vec_create((size_t)A->M);
            if (!y.data)
                die("y allocation", -ENOMEM);
            for (int r = 0; r < opt.gpus; ++r) {
                rc = spmv_mgpu_get_y(g, r, y.data);
                if (rc)
                    break;
                check(fmt ? "multi-GPU HLL" : "multi-GPU CSR", y);
            }
            vec_put(&y);
        }
        int64_t nz = 0, bytes = 0;
        spmv_mgpu_info(g, NULL, NULL, &nz, NULL);
        for (int r = 0; r < opt.gpus; ++r) { /* shards may differ in size */
            int64_t b = 0;
            if (spmv_mgpu_shard_info(g, r, NULL, &b, NULL, 0) == 0)
                bytes += b;
        }
        double med = median_of(ms, opt.iters > 0 ? opt.iters : 1);
        log_roofline(A->name, fmt ? "HLL" : "CSR", fmt ? 1 : 2, 8, opt.gpus,
                     A->M, A->N, nz, nz, bytes, med);
        LOG_INFO("%d GPU(s) %s: %.4f ms per step (kernel + all-gather), "
                 "%.1f GFLOP/s", opt.gpus, fmt ? "HLL" : "CSR", med,
                 compute_gflops64(med, nz));
    }
    free(ms);
    spmv_mgpu_destroy(g);
    if (rc)
        die("multi-GPU run failed", rc);
}

static int synth_kind(const char *s) {
    static const char *names[] = {"banded", "random", "ragged", "kkt",
                                  "stencil", "powerlaw", "hub"};
    for (int k = 0; k < 7; ++k)
        if (!strcmp(s, names[k]))
            return k;
    return -1;
}

int main(int argc, char **argv) {
    static const struct option longopts[] = {
        {"matrix", required_argument, NULL, 'm'},
        {"out", required_argument, NULL, 'o'},
        {"synthetic", required_argument, NULL, 's'},
        {"rows", required_argument, NULL, 'R'},
        {"nnz-row", required_argument, NULL, 'K'},
        {"window", required_argument, NULL, 'W'},
        {"iters", required_argument, NULL, 'i'},
        {"gpus", required_argument, NULL, 'g'},
        {"only-multi-gpu", no_argument, NULL, 1003},
        {"partition", required_argument, NULL, 1004},
        {"ragged-exchange", required_argument, NULL, 1005},
        {"exchange-chunks", required_argument, NULL, 1006},
        {"logical-shards", required_argument, NULL, 1007},
        {"exchange-engine", required_argument, NULL, 1008},
        {"no-cpu", no_argument, NULL, 'C'},
        {"debug", no_argument, NULL, 'd'},
        {"help", no_argument, NULL, 'h'},
        {NULL, 0, NULL, 0}};
    int c;
    while ((c = getopt_long(argc, argv, "m:o:s:i:g:dh", longopts, NULL)) != -1) {
        switch (c) {
        case 'm': opt.matrix = optarg; break;
        case 'o': opt.out_dir = optarg; break;
        case 's': opt.synthetic = optarg; break;
        case 'R': opt.rows = atoi(optarg); break;
        case 'K': opt.nnz_row = atoi(optarg); break;
        case 'W': opt.window = atoll(optarg); break;
        case 'i': opt.iters = atoi(optarg); break;
        case 'g': opt.gpus = atoi(optarg); break;
        case 1003: opt.only_mgpu = 1; break;
        case 1004:
            if (!strcmp(optarg, "nnz"))
                opt.partition = SPMV_MGPU_PART_NNZ;
            else if (!strcmp(optarg, "even"))
                opt.partition = SPMV_MGPU_PART_EVEN;
            else {
                LOG_ERR("--partition takes even or nnz, not %s", optarg);
                return EXIT_FAILURE;
            }
            break;
        case 1005:
            if (!strcmp(optarg, "p2p"))
                opt.xchg = SPMV_MGPU_XCHG_P2P;
            else if (!strcmp(optarg, "bcast"))
                opt.xchg = SPMV_MGPU_XCHG_BCAST;
            else if (!strcmp(optarg, "padded"))
                opt.xchg = SPMV_MGPU_XCHG_PADDED;
            else {
                LOG_ERR("--ragged-exchange takes p2p, bcast or padded, not %s",
                        optarg);
                return EXIT_FAILURE;
            }
            break;
        case 1006:
            opt.chunks = atoi(optarg);
            if (opt.chunks < 1 || opt.chunks > 16) {
                LOG_ERR("--exchange-chunks takes 1..16");
                return EXIT_FAILURE;
            }
            break;
        case 1007:
            opt.shards = atoi(optarg);
            if (opt.shards < 1 || opt.shards > 16) {
                LOG_ERR("--logical-shards takes 1..16");
                return EXIT_FAILURE;
            }
            break;
        case 1008:
            if (!strcmp(optarg, "copy"))
                opt.engine = SPMV_MGPU_ENGINE_COPY;
            else if (!strcmp(optarg, "rccl"))
                opt.engine = SPMV_MGPU_ENGINE_RCCL;
            else {
                LOG_ERR("--exchange-engine takes rccl or copy, not %s", optarg);
                return EXIT_FAILURE;
            }
            break;
        case 'C': opt.cpu = 0; break;
        case 'd': opt.debug = true; break;
        case 'h':
            log_prog_usage(basename(argv[0]));
            return EXIT_SUCCESS;
        default:
            log_prog_usage(basename(argv[0]));
            return EXIT_FAILURE;
        }
    }
    if ((!opt.matrix && !opt.synthetic) || !opt.out_dir) {
        log_prog_usage(basename(argv[0]));
        return EXIT_FAILURE;
    }
    if (opt.gpus < 1 || opt.gpus > 64) {
        LOG_ERR("--gpus takes 1..64");
        return EXIT_FAILURE;
    }
    if (logger_init(opt.out_dir)) {
        LOG_ERR("Failed to open log files in: %s", opt.out_dir);
        return EXIT_FAILURE;
    }
    logger_open = true;

    if (opt.matrix) {
        A = io_load_csr(opt.matrix);
    } else {
        int kind = synth_kind(opt.synthetic);
        if (kind < 0)
            die("unknown synthetic family", -EINVAL);
        long long w = opt.window > 0 ? opt.window : 2ll * opt.rows;
        A = csr_generate(kind, opt.rows, opt.rows, opt.nnz_row, w, 0, 42);
    }
    if (IS_ERR(A)) {
        int code = PTR_ERR(A);
        A = NULL;
        die(opt.matrix ? opt.matrix : opt.synthetic, code);
    }
    H_row = csr_to_hll(A, false);
    H_col = csr_to_hll(A, true);
    if (IS_ERR(H_row) || IS_ERR(H_col))
        die("CSR -> HLL conversion failed", -ENOMEM);

    x = vec_create((size_t)A->N);
    if (!x.data)
        die("x allocation failed", -ENOMEM);
    if (opt.matrix)
        vec_fill_random(&x); /* the reference's x (main.c:97-102) */
    else
        vec_fill_synth(&x, 7, 0);

    if (opt.only_mgpu) /* a GPU-count sweep re-runs only the -g N step */
        opt.cpu = 0;
    if (opt.cpu || opt.debug)
        run_serial();
    if (opt.cpu)
        run_omp();
    if (spmv_device_count() > 0) {
        if (!opt.only_mgpu)
            run_gpu();
        if (opt.gpus >= 1 &&
            (opt.gpus > 1 || opt.only_mgpu || getenv("SPMV_FORCE_MGPU")))
            run_multi_gpu();
    } else
        LOG_WARN("no GPU visible: GPU benchmarks skipped (no CPU fallback)");

    cleanup();
    return EXIT_SUCCESS;
}

/*
 * ref_harness.c -- drives the REFERENCE's own CPU code (test infrastructure).
 *
 * This file is compiled together with the reference's C sources where they
 * lie under /root/reference/src (csr.c, hll.c, vector.c, utils.c, mmio.c) by
 * oracle/build_ref.sh; the result goes to oracle/_ref/ (git-ignored).  No
 * reference source is copied and nothing is stubbed: the CUDA entry points
 * that csr.c/hll.c mention are only reachable from bench_*_cuda_* wrappers,
 * which this harness never calls, so the linker drops those sections
 * (-ffunction-sections -Wl,--gc-sections) together with their undefined
 * symbols.
 *
 * Uses only the reference's public API (include/csr.h, hll.h, vector.h,
 * utils.h, err.h):
 *   dump  <file.mtx>         io_load_csr -> csr_to_hll(row,col) -> x from
 *                            vec_fill_random -> bench_csr_serial /
 *                            bench_hll_serial; prints every array.
 *   err   <file.mtx>         prints the loader's error code.
 *   synth <kind> <M> <N> <K> <W> <seed> <xseed> <nsample>
 *                            fills a sparse_csr from include/spmv_synth.h,
 *                            runs the reference serial CSR + HLL kernels and
 *                            prints a strided sample of y + a checksum.
 *   time  <kind> <M> <N> <K> <W> <seed> <xseed> <reps> <thr> [<thr>...]
 *                            CPU baseline: reference serial + OpenMP benches
 *                            timed on the host cores (JSON on stdout);
 *                            REF_TIME_WINDOW_MS: every sample repeats the
 *                            single-shot bench until that much time is
 *                            covered (CFS quota: see cmd_time).
 *
 * Doubles are printed as C99 hex floats (%a): exact round trip.
 */
#include <errno.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "csr.h"
#include "err.h"
#include "hll.h"
#include "utils.h"
#include "vector.h"

#include "../include/spmv_synth.h"

static void put_ints(const char *key, const int *v, long n) {
    printf("%s %ld", key, n);
    for (long i = 0; i < n; ++i)
        printf(" %d", v[i]);
    printf("\n");
}

static void put_dbls(const char *key, const double *v, long n) {
    printf("%s %ld", key, n);
    for (long i = 0; i < n; ++i)
        printf(" %a", v[i]);
    printf("\n");
}

static void put_hll(const char *tag, const sparse_hll *H) {
    printf("%s.hdr 5 %d %d %d %d %d\n", tag, H->M, H->N, H->NZ, H->hack_size,
           H->num_blocks);
    for (int b = 0; b < H->num_blocks; ++b) {
        const ellpack_block *blk = &H->blocks[b];
        char key[64];
        long len = (long)blk->M * blk->max_NZ;
        printf("%s.blk%d 4 %d %d %d %d\n", tag, b, blk->M, blk->N, blk->NZ,
               blk->max_NZ);
        snprintf(key, sizeof key, "%s.blk%d.JA", tag, b);
        put_ints(key, blk->JA, len);
        snprintf(key, sizeof key, "%s.blk%d.AS", tag, b);
        put_dbls(key, blk->AS, len);
    }
}

static int cmd_dump(const char *path) {
    sparse_csr *A = io_load_csr(path);
    if (IS_ERR(A)) {
        printf("error 1 %d\n", PTR_ERR(A));
        return 0;
    }
    printf("name %s\n", A->name);
    printf("shape 3 %d %d %d\n", A->M, A->N, A->NZ);
    put_ints("IRP", A->IRP, (long)A->M + 1);
    put_ints("JA", A->JA, A->NZ);
    put_dbls("AS", A->AS, A->NZ);

    sparse_hll *Hr = csr_to_hll(A, false);
    sparse_hll *Hc = csr_to_hll(A, true);
    if (IS_ERR(Hr) || IS_ERR(Hc)) {
        printf("error 1 %d\n", -ENOMEM);
        return 0;
    }
    put_hll("hll_row", Hr);
    put_hll("hll_col", Hc);

    vec x = vec_create((size_t)A->N);
    vec_fill_random(&x); /* glibc rand(), never seeded == srand(1) */
    put_dbls("x", x.data, (long)x.len);

    bench rc, rh;
    if (bench_csr_serial(A, x.data, &rc) || bench_hll_serial(Hr, x.data, &rh))
        return 2;
    put_dbls("y_csr_serial", rc.data.data, (long)rc.data.len);
    put_dbls("y_hll_serial", rh.data.data, (long)rh.data.len);
    printf("validate 1 %d\n", validation_vec_result(rc.data, rh.data));
    printf("gflops_probe 1 %a\n", compute_gflops(2.0, A->NZ));

    bench_omp og = {.num_threads = 2}, on = {.num_threads = 2},
              oh = {.num_threads = 2};
    if (bench_csr_omp_guided(A, x.data, &og) ||
        bench_csr_omp_nnz_balancing(A, x.data, &on) ||
        bench_hll_omp(Hr, x.data, &oh))
        return 2;
    put_dbls("y_csr_omp_guided", og.bench.data.data, (long)og.bench.data.len);
    put_dbls("y_csr_omp_nnz", on.bench.data.data, (long)on.bench.data.len);
    put_dbls("y_hll_omp", oh.bench.data.data, (long)oh.bench.data.len);
    printf("omp_names %s %s %s\n", og.name, on.name, oh.name);
    printf("omp_nnz_threads 1 %d\n", on.num_threads);

    vec_put(&rc.data);
    vec_put(&rh.data);
    vec_put(&og.bench.data);
    vec_put(&on.bench.data);
    vec_put(&oh.bench.data);
    vec_put(&x);
    hll_free(Hr);
    hll_free(Hc);
    csr_free(A);
    return 0;
}

static int cmd_err(const char *path) {
    sparse_csr *A = io_load_csr(path);
    if (IS_ERR(A)) {
        printf("error 1 %d\n", PTR_ERR(A));
    } else {
        printf("error 1 0\n");
        csr_free(A);
    }
    return 0;
}

/* Build a reference sparse_csr from the shared synthetic definition. */
static sparse_csr *synth_csr(const synth_spec *s) {
    int M = s->M;
    int *IRP = aligned_malloc(((size_t)M + 1) * sizeof(int));
    if (!IRP)
        return NULL;
    long nz = 0;
    IRP[0] = 0;
    for (int i = 0; i < M; ++i) {
        nz += synth_row_len(s, s->row0 + i);
        IRP[i + 1] = (int)nz;
    }
    int *JA = aligned_malloc((size_t)nz * sizeof(int));
    double *AS = aligned_malloc((size_t)nz * sizeof(double));
    sparse_csr *A = malloc(sizeof *A);
    if (!JA || !AS || !A)
        return NULL;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < M; ++i)
        synth_fill_row(s, s->row0 + i, IRP[i + 1] - IRP[i], JA + IRP[i],
                       AS + IRP[i]);
    init_csr(A, "synthetic", M, s->N, (int)nz, IRP, JA, AS);
    return A;
}

static int parse_spec(char **a, synth_spec *s, uint64_t *xseed) {
    s->kind = atoi(a[0]);
    s->M = atoi(a[1]);
    s->N = atoi(a[2]);
    s->K = atoi(a[3]);
    s->W = atoll(a[4]);
    s->seed = strtoull(a[5], NULL, 10);
    s->row0 = 0;
    *xseed = strtoull(a[6], NULL, 10);
    return 0;
}

static int cmd_synth(char **a) {
    synth_spec s;
    uint64_t xseed;
    parse_spec(a, &s, &xseed);
    int nsample = atoi(a[7]);
    sparse_csr *A = synth_csr(&s);
    if (!A)
        return 2;
    vec x = vec_create((size_t)s.N);
    for (int i = 0; i < s.N; ++i)
        x.data[i] = synth_x(xseed, i);
    sparse_hll *H = csr_to_hll(A, false);
    bench rc, rh;
    if (IS_ERR(H) || bench_csr_serial(A, x.data, &rc) ||
        bench_hll_serial(H, x.data, &rh))
        return 2;
    printf("shape 3 %d %d %d\n", A->M, A->N, A->NZ);
    long stride = nsample > 0 ? (A->M + nsample - 1) / nsample : 1;
    if (stride < 1)
        stride = 1;
    printf("stride 1 %ld\n", stride);
    printf("y_csr_serial_sample");
    long cnt = 0;
    for (long i = 0; i < A->M; i += stride)
        ++cnt;
    printf(" %ld", cnt);
    for (long i = 0; i < A->M; i += stride)
        printf(" %a", rc.data.data[i]);
    printf("\n");
    double sum = 0.0, asum = 0.0;
    int same = 1;
    for (int i = 0; i < A->M; ++i) {
        sum += rc.data.data[i];
        asum += rc.data.data[i] < 0 ? -rc.data.data[i] : rc.data.data[i];
        if (rc.data.data[i] != rh.data.data[i])
            same = 0;
    }
    printf("y_sum 1 %a\n", sum);
    printf("y_abs_sum 1 %a\n", asum);
    printf("hll_bit_equal 1 %d\n", same);
    return 0;
}

static double wall_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

static int cmp_d(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

static double median(double *v, int n) {
    qsort(v, (size_t)n, sizeof *v, cmp_d);
    return (n & 1) ? v[n / 2] : 0.5 * (v[n / 2 - 1] + v[n / 2]);
}

/*
 * CPU baseline: the reference's own serial and OpenMP benches, each call a
 * single shot as the reference times it (clock() for serial, omp_get_wtime()
 * for OpenMP).  One SAMPLE = calls repeated until their reported durations add
 * up to REF_TIME_WINDOW_MS (default 0: one call), its value the mean duration
 * per call; `reps` samples per leg; we report the first call (cold, the
 * reference's own protocol) and the MEDIAN of the samples.  The window exists
 * because of CFS bandwidth control: under a cgroup CPU quota a run shorter
 * than one 100 ms period can burn more CPU than the quota sustains (round 3:
 * a 44 ms HLL run at 40 threads under a 16-CPU quota read 14.4 GFLOP/s, twice
 * what the same box sustains); a window of >= 3 periods cannot.
 */
typedef struct leg_times {
    double cold_ms, median_ms;
    int calls; /* per sample, largest */
} leg_times;

#define LEG(out_, reps_, window_ms_, decl_, call_, dur_, put_)                 \
    do {                                                                       \
        double smp_[64];                                                       \
        (out_).calls = 0;                                                      \
        for (int r_ = 0; r_ < (reps_); ++r_) {                                 \
            double acc_ = 0.0;                                                 \
            int n_ = 0;                                                        \
            do {                                                               \
                decl_;                                                         \
                if (call_)                                                     \
                    return 2;                                                  \
                if (r_ == 0 && n_ == 0)                                        \
                    (out_).cold_ms = (dur_);                                   \
                acc_ += (dur_);                                                \
                ++n_;                                                          \
                put_;                                                          \
            } while (acc_ < (window_ms_) && n_ < 4096);                        \
            smp_[r_] = acc_ / n_;                                              \
            if (n_ > (out_).calls)                                             \
                (out_).calls = n_;                                             \
        }                                                                      \
        (out_).median_ms = median(smp_, (reps_));                              \
    } while (0)

static int cmd_time(int argc, char **a) {
    synth_spec s;
    uint64_t xseed;
    parse_spec(a, &s, &xseed);
    int reps = atoi(a[7]);
    if (reps < 1)
        reps = 1;
    if (reps > 64)
        reps = 64;
    const char *ew = getenv("REF_TIME_WINDOW_MS");
    const double window_ms = ew ? atof(ew) : 0.0;
    double t0 = wall_ms();
    sparse_csr *A = synth_csr(&s);
    if (!A)
        return 2;
    vec x = vec_create((size_t)s.N);
    for (int i = 0; i < s.N; ++i)
        x.data[i] = synth_x(xseed, i);
    /* REF_TIME_HLL=0 skips the HLL legs: the reference's csr_to_hll is a
     * serial loop with two mallocs per hack block (hll.c:19-95), ~15 s for
     * the 10M x 32 matrix.  REF_TIME_HLL=best (what bench.py asks for) keeps
     * them bounded instead: ONE conversion after the CSR ladder, then
     * bench_hll_serial and bench_hll_omp at the thread count that was best
     * for CSR, `reps` samples each like every other leg (hll.c:127-150,
     * 178-211).  REF_TIME_HLL=ladder does the same conversion after the CSR
     * ladder and then runs bench_hll_omp at EVERY thread count of the ladder,
     * as the reference's driver does (main.c:176-253), for as long as the HLL
     * legs stay inside REF_TIME_HLL_BUDGET_MS (default 15000); the counts
     * that did not fit are listed in "hll_skipped_threads". */
    const char *eh = getenv("REF_TIME_HLL");
    const int hll_ladder = eh && !strcmp(eh, "ladder");
    const int hll_best = hll_ladder || (eh && !strcmp(eh, "best"));
    const char *eb = getenv("REF_TIME_HLL_BUDGET_MS");
    const double hll_budget_ms = eb ? atof(eb) : 15000.0;
    const int with_hll = !hll_best && !(eh && eh[0] == '0');
    sparse_hll *H = with_hll ? csr_to_hll(A, false) : NULL;
    if (with_hll && IS_ERR(H))
        return 2;
    double t_prep = wall_ms() - t0;
    leg_times lt;

    printf("{\"rows\": %d, \"cols\": %d, \"nnz\": %d, \"prep_ms\": %.3f, "
           "\"omp_max_threads\": %d, \"procs\": %d, \"reps\": %d, "
           "\"window_ms\": %.1f, \"runs\": [",
           A->M, A->N, A->NZ, t_prep, omp_get_max_threads(),
           omp_get_num_procs(), reps, window_ms);
    int first = 1;
#define EMIT(fmt_, bench_, thr_)                                               \
    do {                                                                       \
        printf("%s{\"format\": \"%s\", \"bench\": \"%s\", \"threads\": %d, "   \
               "\"cold_ms\": %.6f, \"median_ms\": %.6f, \"gflops\": %.6f, "    \
               "\"calls_per_sample\": %d}",                                    \
               first ? "" : ", ", fmt_, bench_, thr_, lt.cold_ms,              \
               lt.median_ms, compute_gflops(lt.median_ms, A->NZ), lt.calls);   \
        first = 0;                                                             \
    } while (0)

    LEG(lt, reps, window_ms, bench b, bench_csr_serial(A, x.data, &b),
        b.duration_ms, vec_put(&b.data));
    EMIT("CSR", "serial", 1);
    if (with_hll) {
        LEG(lt, reps, window_ms, bench b, bench_hll_serial(H, x.data, &b),
            b.duration_ms, vec_put(&b.data));
        EMIT("HLL", "serial", 1);
    }

    int best_thr = 1;
    double best_csr_ms = 1e300;
    for (int k = 8; k < argc; ++k) {
        int thr = atoi(a[k]);
        if (thr < 1 || thr > omp_get_max_threads())
            continue; /* reference asserts on this (hll.c:184, csr.c:320) */
        OMP_WARMUP(thr);
        LEG(lt, reps, window_ms, bench_omp b = {.num_threads = thr},
            bench_csr_omp_guided(A, x.data, &b), b.bench.duration_ms,
            vec_put(&b.bench.data));
        if (lt.median_ms < best_csr_ms) {
            best_csr_ms = lt.median_ms;
            best_thr = thr;
        }
        EMIT("CSR", "omp_guided", thr);
        int used = thr;
        LEG(lt, reps, window_ms, bench_omp b = {.num_threads = thr},
            bench_csr_omp_nnz_balancing(A, x.data, &b), b.bench.duration_ms,
            (used = b.num_threads, vec_put(&b.bench.data)));
        if (lt.median_ms < best_csr_ms) {
            best_csr_ms = lt.median_ms;
            best_thr = thr;
        }
        EMIT("CSR", "omp_nnz", used);
        if (with_hll) {
            LEG(lt, reps, window_ms, bench_omp b = {.num_threads = thr},
                bench_hll_omp(H, x.data, &b), b.bench.duration_ms,
                vec_put(&b.bench.data));
            EMIT("HLL", "omp_guided", thr);
        }
    }
    double hll_prep = 0.0;
    int skipped[16], nskip = 0;
    if (hll_best) {
        double t1 = wall_ms();
        H = csr_to_hll(A, false);
        if (IS_ERR(H))
            return 2;
        hll_prep = wall_ms() - t1;
        LEG(lt, reps, window_ms, bench b, bench_hll_serial(H, x.data, &b),
            b.duration_ms, vec_put(&b.data));
        EMIT("HLL", "serial", 1);
        if (hll_ladder) {
            const double t2 = wall_ms();
            for (int k = 8; k < argc; ++k) {
                const int thr = atoi(a[k]);
                if (thr < 2 || thr > omp_get_max_threads())
                    continue;
                if (wall_ms() - t2 > hll_budget_ms) { /* out of the time box */
                    if (nskip < 16)
                        skipped[nskip++] = thr;
                    continue;
                }
                OMP_WARMUP(thr);
                LEG(lt, reps, window_ms, bench_omp bo = {.num_threads = thr},
                    bench_hll_omp(H, x.data, &bo), bo.bench.duration_ms,
                    vec_put(&bo.bench.data));
                EMIT("HLL", "omp_guided", thr);
            }
        } else if (best_thr > 1) {
            OMP_WARMUP(best_thr);
            LEG(lt, reps, window_ms, bench_omp bo = {.num_threads = best_thr},
                bench_hll_omp(H, x.data, &bo), bo.bench.duration_ms,
                vec_put(&bo.bench.data));
            EMIT("HLL", "omp_guided", best_thr);
        }
    }
    printf("], \"hll_prep_ms\": %.3f, \"hll_blocks\": %d, "
           "\"hll_skipped_threads\": [", hll_prep, H ? H->num_blocks : 0);
    for (int k = 0; k < nskip; ++k)
        printf("%s%d", k ? ", " : "", skipped[k]);
    printf("]}\n");
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 3 && !strcmp(argv[1], "dump"))
        return cmd_dump(argv[2]);
    if (argc >= 3 && !strcmp(argv[1], "err"))
        return cmd_err(argv[2]);
    if (argc >= 10 && !strcmp(argv[1], "synth"))
        return cmd_synth(argv + 2);
    if (argc >= 10 && !strcmp(argv[1], "time"))
        return cmd_time(argc - 2, argv + 2);
    fprintf(stderr, "usage: %s dump|err <file.mtx> | synth ... | time ...\n",
            argv[0]);
    return 64;
}

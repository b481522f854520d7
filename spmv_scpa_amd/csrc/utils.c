/* utils.c -- allocation, validation and usage text (API: include/utils.h). */
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>

#include <string.h>
#include "utils.h"

void *aligned_malloc(size_t size) {
    void *p = NULL;
    if (posix_memalign(&p, ALIGNMENT, size ? size : ALIGNMENT) != 0)
        return NULL;
    return p;
}

/* The reference's -d criterion (utils.c:39-60): absolute L2 distance. */
int validation_vec_result(const vec expected, const vec res) {
    if (expected.len != res.len)
        return -1;
    double acc = 0.0;
    for (size_t i = 0; i < res.len; ++i) {
        double d = expected.data[i] - res.data[i];
        acc += d * d;
    }
    return sqrt(acc) > 1e-1 ? -1 : 0;
}

double max_rel_err(const vec expected, const vec res, const double *scale) {
    if (expected.len != res.len)
        return -1.0;
    double worst = 0.0;
    for (size_t i = 0; i < res.len; ++i) {
        double den = fabs(expected.data[i]);
        double fl = scale ? 1e-3 * scale[i] : 0.0;
        if (fl > den)
            den = fl;
        if (den < 1e-300)
            den = 1e-300;
        double e = fabs(res.data[i] - expected.data[i]) / den;
        if (e > worst || e != e)
            worst = (e != e) ? INFINITY : e;
    }
    return worst;
}

void log_prog_usage(const char *prog) {
    fprintf(stderr,
            "Usage: %s (-m <matrix.mtx> | -s <family>) -o <out-dir> [options]\n"
            "  -m, --matrix <file>     Matrix Market file to process\n"
            "  -s, --synthetic <kind>  banded | random | ragged | kkt | stencil |\n"
            "                          powerlaw | hub\n"
            "      --rows <M> --nnz-row <K> --window <W>   synthetic shape\n"
            "  -o, --out <dir>         directory for serial.csv omp.csv cuda.csv\n"
            "  -d, --debug             validate every result against serial CSR\n"
            "  -g, --gpus <n>          row-partition over n GPUs + RCCL all-gather(y)\n"
            "      --partition <p>     even (rows, default) | nnz (entries per GPU\n"
            "                          balanced; ragged y fragments)\n"
            "      --ragged-exchange <x>  p2p (default) | bcast | padded\n"
            "      --exchange-chunks <k>  opt-in staged all-gather, k row chunks\n"
            "      --exchange-engine <e>  rccl (collectives, default) | copy (peer\n"
            "                          copies on the copy engines)\n"
            "      --logical-shards <L>   opt-in: L matrices per GPU, the all-gather\n"
            "                          of shard c under the kernel of c+1 (also for\n"
            "                          the blocked path)\n"
            "  -i, --iters <n>         timed GPU launches per kernel (default 20)\n"
            "      --no-cpu            skip the serial / OpenMP benchmarks\n"
            "      --only-multi-gpu    run only the -g <n> step (GPU-count sweeps)\n"
            "  -h, --help              show this message\n",
            prog);
}

void print_result_vector(const vec res) {
    printf("Result vector y (length %zu)\n", res.len);
    for (size_t i = 0; i < res.len; ++i)
        printf("  y[%zu] = %.4f\n", i, res.data[i]);
    printf("\n");
}

/*
 * Team size of the library's OWN host loops (loader, generators, converters,
 * packing): libgomp sizes a team from the affinity mask -- 256 hardware
 * threads on the MI355X boxes -- not from the cgroup CPU quota (16 CPUs
 * there), and a 256-thread team on 16 CPUs of quota spends its time being
 * throttled (a 240 MB packing loop: 183 ms with the default team, 60 ms with
 * 8 threads).  min(omp_get_max_threads(), cgroup v2 quota), cached.  The CPU
 * BENCHMARKS are not affected: they run the thread counts they are asked for.
 */
int spmv_host_threads(void) {
    static int cached;
    int t = __atomic_load_n(&cached, __ATOMIC_RELAXED);
    if (t > 0)
        return t;
    t = omp_get_max_threads();
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
        char q[32];
        long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") &&
            period > 0) {
            const long quota = atol(q);
            const int cpus = (int)((quota + period - 1) / period);
            if (cpus >= 1 && cpus < t)
                t = cpus;
        }
        fclose(f);
    }
    if (t < 1)
        t = 1;
    __atomic_store_n(&cached, t, __ATOMIC_RELAXED);
    return t;
}

void omp_warmup(int num_threads) {
    if (num_threads < 1)
        num_threads = 1;
    double sink = 0.0;
#pragma omp parallel for schedule(guided) num_threads(num_threads) reduction(+ : sink)
    for (int j = 0; j < 1000000; ++j)
        sink += j * 0.5;
    volatile double keep = sink;
    (void)keep;
}

/*
 * utils.h -- benchmark records, timing and small helpers of the host API.
 *
 * Field-for-field compatible with the reference's utils.h:32-47 records
 * (`bench`, `bench_omp`, and the GPU record with its per-launch knob) and
 * its GFLOP/s definition (utils.h:70-75): 2*nnz / (ms * 1e6), 0 when the
 * duration is not positive.
 */
#ifndef SPMV_UTILS_H
#define SPMV_UTILS_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <time.h>

#include "vector.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MAX_PATH 256
#define MAX_NAME 64
#define ALIGNMENT 64 /* bytes; every matrix/vector array */
#define ARRAY_SIZE(a) (sizeof(a) / sizeof((a)[0]))

/* One timed run.  `data` is y; ownership passes to the caller (vec_put). */
typedef struct benchmark_result {
    double duration_ms;
    double gflops;
    vec data;
} bench;

typedef struct benchmark_omp {
    bench bench;
    char name[MAX_NAME]; /* "omp_guided" | "omp_nnz" */
    int num_threads;     /* in: requested; out: used (nnz balancing may shrink) */
} bench_omp;

/* GPU run.  `waves_per_block` is the launch knob the reference calls
 * warps_per_block (utils.h:44-47); here a wave is 64 lanes. */
typedef struct benchmark_hip {
    bench bench;
    int waves_per_block;
} bench_hip;

#define LOG_INFO(fmt, ...)                                                    \
    fprintf(stdout, "[INFO ] %s:%d: " fmt "\n", __FILE__, __LINE__,           \
            ##__VA_ARGS__)
#define LOG_WARN(fmt, ...)                                                    \
    fprintf(stdout, "[WARN ] %s:%d: " fmt "\n", __FILE__, __LINE__,           \
            ##__VA_ARGS__)

/* CPU time of this process in ms (reference utils.h:68 uses clock()). */
static inline double now(void) {
    return (double)clock() * 1e3 / (double)CLOCKS_PER_SEC;
}

/* Monotonic wall clock in ms. */
static inline double wall_now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec * 1e3 + (double)t.tv_nsec * 1e-6;
}

static inline double compute_gflops64(double duration_ms, int64_t nnz) {
    return duration_ms > 0.0 ? (2.0 * (double)nnz) / (duration_ms * 1e6) : 0.0;
}

static inline double compute_gflops(double duration_ms, int nnz) {
    return compute_gflops64(duration_ms, (int64_t)nnz);
}

/* posix_memalign(ALIGNMENT); NULL on failure.  size 0 still yields a
 * unique pointer that free() accepts. */
void *aligned_malloc(size_t size);
/* team size of the library's own host loops: min(OpenMP's default, the
 * cgroup CPU quota) -- see utils.c */
int spmv_host_threads(void);

/* 0 when ||expected - res||_2 <= 0.1 (the reference's -d check,
 * utils.c:39-60); -1 on length mismatch or a larger distance. */
int validation_vec_result(const vec expected, const vec res);

/*
 * Parity metric of this build, stricter than the above:
 * returns max_i |res_i - expected_i| / max(|expected_i|, floor_i) where
 * floor_i = 1e-3 * scale_i (scale = sum_j |a_ij x_j|, may be NULL -> 0)
 * and tiny 1e-300 guards 0/0; -1.0 on length mismatch.
 */
double max_rel_err(const vec expected, const vec res, const double *scale);

void log_prog_usage(const char *prog);
void print_result_vector(const vec res);

/* Spin `num_threads` OpenMP threads up before a timed region. */
void omp_warmup(int num_threads);
#define OMP_WARMUP(n) omp_warmup(n)

#ifdef __cplusplus
}
#endif
#endif /* SPMV_UTILS_H */

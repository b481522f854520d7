/* logger.c -- CSV result files (API and schemas: include/logger.h).
 * Column sets and append semantics follow reference src/logger.c:19-153. */
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>

#include "err.h"
#include "logger.h"

enum { F_SERIAL, F_OMP, F_GPU, F_ROOF, F_COUNT };

static FILE *files[F_COUNT];

static const char *const file_names[F_COUNT] = {"serial.csv", "omp.csv",
                                                "cuda.csv", "roofline.csv"};

static const char *const headers[F_COUNT] = {
    "matrix,format,rows,cols,nnz,num_blocks,duration_ms,gflops\n",
    "matrix,format,bench,rows,cols,nnz,num_blocks,num_threads,duration_ms,"
    "gflops\n",
    "matrix,format,kernel,warps_per_block,rows,cols,nnz,num_blocks,"
    "duration_ms,gflops\n",
    "matrix,format,kernel,waves_per_block,gpus,rows,cols,nnz,slots,bytes,"
    "duration_ms,gflops,gbps,roofline_frac\n"};

/* append; the header goes in only when the file did not exist before */
static FILE *open_csv(const char *dir, int which) {
    char path[MAX_PATH + 32];
    struct stat st;
    snprintf(path, sizeof path, "%s/%s", dir, file_names[which]);
    int fresh = stat(path, &st) != 0;
    FILE *f = fopen(path, "a");
    if (f && fresh) {
        fputs(headers[which], f);
        fflush(f);
    }
    return f;
}

int logger_init(const char *base_path) {
    int ok = 1;
    for (int k = 0; k < F_COUNT; ++k) {
        files[k] = open_csv(base_path, k);
        ok &= files[k] != NULL;
    }
    if (!ok) {
        logger_close();
        return -1;
    }
    return 0;
}

void logger_close(void) {
    for (int k = 0; k < F_COUNT; ++k) {
        if (files[k])
            fclose(files[k]);
        files[k] = NULL;
    }
}

static FILE *sink(int which) {
    if (!files[which])
        LOG_ERR("%s is not open (logger_init not called?)", file_names[which]);
    return files[which];
}

void log_csr_serial_benchmark(const sparse_csr *A, bench r) {
    FILE *f = sink(F_SERIAL);
    if (!f)
        return;
    fprintf(f, "%s,CSR,%d,%d,%d,,%f,%f\n", A->name, A->M, A->N, A->NZ,
            r.duration_ms, r.gflops);
    fflush(f);
}

void log_hll_serial_benchmark(const sparse_hll *H, bench r) {
    FILE *f = sink(F_SERIAL);
    if (!f)
        return;
    fprintf(f, "%s,HLL,%d,%d,%d,%d,%f,%f\n", H->name, H->M, H->N, H->NZ,
            H->num_blocks, r.duration_ms, r.gflops);
    fflush(f);
}

void log_csr_omp_benchmark(const sparse_csr *A, bench_omp r) {
    FILE *f = sink(F_OMP);
    if (!f)
        return;
    fprintf(f, "%s,CSR,%s,%d,%d,%d,,%d,%f,%f\n", A->name, r.name, A->M, A->N,
            A->NZ, r.num_threads, r.bench.duration_ms, r.bench.gflops);
    fflush(f);
}

void log_hll_omp_benchmark(const sparse_hll *H, bench_omp r) {
    FILE *f = sink(F_OMP);
    if (!f)
        return;
    fprintf(f, "%s,HLL,%s,%d,%d,%d,%d,%d,%f,%f\n", H->name, r.name, H->M, H->N,
            H->NZ, H->num_blocks, r.num_threads, r.bench.duration_ms,
            r.bench.gflops);
    fflush(f);
}

void log_csr_hip_benchmark(const sparse_csr *A, bench_hip r, int kernel_id) {
    FILE *f = sink(F_GPU);
    if (!f)
        return;
    fprintf(f, "%s,CSR,%d,%d,%d,%d,%d,,%f,%f\n", A->name, kernel_id,
            r.waves_per_block, A->M, A->N, A->NZ, r.bench.duration_ms,
            r.bench.gflops);
    fflush(f);
}

void log_hll_hip_benchmark(const sparse_hll *H, bench_hip r, int kernel_id) {
    FILE *f = sink(F_GPU);
    if (!f)
        return;
    fprintf(f, "%s,HLL,%d,%d,%d,%d,%d,%d,%f,%f\n", H->name, kernel_id,
            r.waves_per_block, H->M, H->N, H->NZ, H->num_blocks,
            r.bench.duration_ms, r.bench.gflops);
    fflush(f);
}

void log_roofline(const char *matrix, const char *format, int kernel_id,
                  int waves_per_block, int gpus, int rows, int cols,
                  int64_t nnz, int64_t slots, int64_t bytes,
                  double duration_ms) {
    FILE *f = sink(F_ROOF);
    if (!f)
        return;
    double gbps = duration_ms > 0 ? (double)bytes / (duration_ms * 1e6) : 0.0;
    fprintf(f, "%s,%s,%d,%d,%d,%d,%d,%lld,%lld,%lld,%f,%f,%f,%f\n", matrix,
            format, kernel_id, waves_per_block, gpus, rows, cols,
            (long long)nnz, (long long)slots, (long long)bytes, duration_ms,
            compute_gflops64(duration_ms, nnz), gbps,
            gbps / (8000.0 * (gpus > 0 ? gpus : 1)));
    fflush(f);
}

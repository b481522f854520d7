/*
 * logger.h -- CSV result files of the driver.
 *
 * Same files, columns and append semantics as the reference
 * (logger.c:19-54, rows logger.c:89-153): a header is written only when the
 * file is created, each row is flushed, numbers use "%f".
 *
 *   serial.csv  matrix,format,rows,cols,nnz,num_blocks,duration_ms,gflops
 *   omp.csv     matrix,format,bench,rows,cols,nnz,num_blocks,num_threads,
 *               duration_ms,gflops
 *   cuda.csv    matrix,format,kernel,warps_per_block,rows,cols,nnz,
 *               num_blocks,duration_ms,gflops         (GPU rows; name kept
 *               so the reference's plotting script reads it unchanged)
 *   roofline.csv (new)  matrix,format,kernel,waves_per_block,gpus,rows,cols,
 *               nnz,slots,bytes,duration_ms,gflops,gbps,roofline_frac
 */
#ifndef SPMV_LOGGER_H
#define SPMV_LOGGER_H

#include "csr.h"
#include "hll.h"
#include "utils.h"

#ifdef __cplusplus
extern "C" {
#endif

int logger_init(const char *base_path); /* 0, or -1 if a file cannot open */
void logger_close(void);

void log_csr_serial_benchmark(const sparse_csr *A, bench res);
void log_hll_serial_benchmark(const sparse_hll *H, bench res);
void log_csr_omp_benchmark(const sparse_csr *A, bench_omp res);
void log_hll_omp_benchmark(const sparse_hll *H, bench_omp res);
void log_csr_hip_benchmark(const sparse_csr *A, bench_hip res, int kernel_id);
void log_hll_hip_benchmark(const sparse_hll *H, bench_hip res, int kernel_id);

/* extra file, one row per GPU measurement with the roofline figures */
void log_roofline(const char *matrix, const char *format, int kernel_id,
                  int waves_per_block, int gpus, int rows, int cols,
                  int64_t nnz, int64_t slots, int64_t bytes,
                  double duration_ms);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_LOGGER_H */

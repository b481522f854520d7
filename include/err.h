/*
 * err.h -- error convention of the host API.
 *
 * Same contract as the reference (include/err.h:10-18): functions that
 * return a pointer encode a small negative errno in the pointer itself;
 * the last 4095 addresses are never valid objects.
 *
 *     sparse_csr *A = io_load_csr(path);
 *     if (IS_ERR(A)) return PTR_ERR(A);      // e.g. -ENOENT, -EINVAL
 *
 * NULL is NOT an error value here (the reference's own driver tests `!A`,
 * main.c:79, and crashes on a missing file: do not copy that).
 */
#ifndef SPMV_ERR_H
#define SPMV_ERR_H

#include <stdint.h>
#include <stdio.h>

#define SPMV_MAX_ERRNO 4095

#define ERR_PTR(code) ((void *)(intptr_t)(code))
#define PTR_ERR(p) ((int)(intptr_t)(p))
#define IS_ERR(p) ((uintptr_t)(p) >= (uintptr_t)(-SPMV_MAX_ERRNO))
#define IS_ERR_OR_NULL(p) (!(p) || IS_ERR(p))

#define LOG_ERR(fmt, ...)                                                     \
    fprintf(stderr, "[ERROR] %s:%d: " fmt "\n", __FILE__, __LINE__,           \
            ##__VA_ARGS__)

#endif /* SPMV_ERR_H */

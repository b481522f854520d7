/*
 * mm_header.h -- the two pieces of Matrix Market parsing the loader needs.
 *
 * The reference vendors NIST mmio (src/mmio.c, 453 lines) but only calls
 * mm_read_banner (mmio.c:93-166) and mm_read_mtx_crd_size (mmio.c:175-200)
 * plus the typecode predicates.  This is an independent implementation of
 * those two behaviours over an in-memory buffer.
 */
#ifndef SPMV_MM_HEADER_H
#define SPMV_MM_HEADER_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mm_info {
    char object;  /* 'M' matrix */
    char format;  /* 'C' coordinate, 'A' array */
    char field;   /* 'R' real, 'C' complex, 'P' pattern, 'I' integer */
    char symmetry;/* 'G' general, 'S' symmetric, 'K' skew, 'H' hermitian */
    int rows, cols, entries;
    size_t data_offset; /* first byte after the size line */
} mm_info;

/*
 * Parse banner + size line from text[0..len).  Returns 0 on success, a
 * positive MM_* code otherwise (any failure makes the loader answer
 * -EINVAL, reference csr.c:48-57).
 */
enum {
    MM_OK = 0,
    MM_PREMATURE_EOF = 12,
    MM_NO_HEADER = 14,
    MM_UNSUPPORTED_TYPE = 15
};
int mm_parse_header(const char *text, size_t len, mm_info *out);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_MM_HEADER_H */

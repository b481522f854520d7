/*
 * vector.h -- dense fp64 vectors of the host API (x and y of y = A*x).
 *
 * Mirrors the reference's vector.h:7-18: `vec` is passed BY VALUE, owns a
 * 64-byte-aligned buffer, and is released with vec_put().
 */
#ifndef SPMV_VECTOR_H
#define SPMV_VECTOR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    size_t len;   /* number of doubles */
    double *data; /* 64 B aligned, or NULL when allocation failed */
} vec;

/* Zero-filled vector of n doubles; .data == NULL on allocation failure
 * (reference vector.c:11-20 memsets before checking: not copied). */
vec vec_create(size_t n);

/* Release the buffer and clear the pointer; safe on NULL / empty. */
void vec_put(vec *v);

void vec_fill(vec *v, double value);

/* x[i] = rand() / RAND_MAX from the C library generator, in index order
 * (reference vector.c:36-41).  The reference never seeds it, so a fresh
 * process gets the srand(1) sequence: 0.8401877171..., 0.3943829268... */
void vec_fill_random(vec *v);

/* Deterministic counter-based fill used by the synthetic workloads:
 * x[i] = synth_x(seed, first + i) of spmv_synth.h. */
void vec_fill_synth(vec *v, uint64_t seed, int64_t first);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_VECTOR_H */

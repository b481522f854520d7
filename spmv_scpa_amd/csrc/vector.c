/* vector.c -- fp64 host vectors (API: include/vector.h). */
#include <stdlib.h>
#include <string.h>

#include "spmv_synth.h"
#include "utils.h"
#include "vector.h"

vec vec_create(size_t n) {
    vec v = {n, NULL};
    v.data = (double *)aligned_malloc(n * sizeof(double));
    if (v.data)
        memset(v.data, 0, n * sizeof(double));
    else
        v.len = 0;
    return v;
}

void vec_put(vec *v) {
    if (!v)
        return;
    free(v->data);
    v->data = NULL;
    v->len = 0;
}

void vec_fill(vec *v, double value) {
    if (!v || !v->data)
        return;
    for (size_t i = 0; i < v->len; ++i)
        v->data[i] = value;
}

/* C library generator, index order: reproduces the reference's x exactly
 * in a process that has not called srand() (reference vector.c:36-41). */
void vec_fill_random(vec *v) {
    if (!v || !v->data)
        return;
    for (size_t i = 0; i < v->len; ++i)
        v->data[i] = (double)rand() / RAND_MAX;
}

void vec_fill_synth(vec *v, uint64_t seed, int64_t first) {
    if (!v || !v->data)
        return;
    int64_t n = (int64_t)v->len;
#pragma omp parallel for schedule(static) if (n > 65536) num_threads(spmv_host_threads())
    for (int64_t i = 0; i < n; ++i)
        v->data[i] = synth_x(seed, first + i);
}

/*
 * spmv_synth.h -- deterministic synthetic sparse matrices and vectors.
 *
 * The reference has no generators: it only reads Matrix Market text
 * (reference src/csr.c:31-171, ~0.6 us per entry).  BASELINE.json's GPU
 * configs are synthetic (1M banded, 10M random), so the workload is
 * DEFINED here, once, as counter-based functions of (seed, row, slot).
 * Every row can be produced independently, on the host, inside the
 * oracle, inside the reference harness and on the device, and all of
 * them get the same bits.  Header only; C99 / C++ / HIP.
 *
 * Matrix families (rows have their columns in ascending order; duplicate
 * columns are kept, as the reference loader keeps them):
 *   SYNTH_BANDED  row i has K entries at columns s..s+K-1,
 *                 s = clamp(i - K/2, 0, N-K).                (config 2)
 *   SYNTH_RANDOM  row i has K entries, columns uniform in the window
 *                 [i - W/2, i + W/2) clipped to [0, N); W >= 2N means
 *                 "anywhere".                                (config 3, 5)
 *   SYNTH_RAGGED  as SYNTH_RANDOM, but the row length is uniform in
 *                 [K - K/4, K + K/4] (mean K): exercises HLL padding.
 *   SYNTH_KKT     irregular: most rows short (K/4..K/2), every 64th row a
 *                 long row of 8K entries, KKT-like arrow structure
 *                 (row-length skew stress test).
 *   SYNTH_STENCIL 3-D grid operator: row g = grid point (ix,iy,iz) of an
 *                 nx x nx x nz grid (nx = W if W > 0, else cbrt(N)), coupled
 *                 to its 27 neighbours (K > 7) or 7 neighbours (K <= 7)
 *                 inside the grid: rows of 8..27 (4..7) entries, banded with
 *                 three band groups -- the structure class of nlpkkt160 /
 *                 FEM matrices (stand-in for config 4, which cannot be
 *                 downloaded here).
 *   SYNTH_POWERLAW web / road / co-purchase graph class (the reference's
 *                 webbase-1M, amazon0302, roadNet-PA, reference
 *                 scripts/download-matrices.py:7-38): row length is a
 *                 discrete power law, P(len >= k) = min(1, (s/k)^1.5) with
 *                 s = (2K+1)/6, i.e. mean ~ K (K = 3: 55 % of the rows hold
 *                 one entry, one row in 10^6 holds >= 10^4, longest 30342),
 *                 computed in integers so that every compiler gets the same
 *                 lengths; columns stratified over the window (slot j of a
 *                 row of L entries is uniform in the j-th of L equal strata
 *                 -- ascending by construction, O(1) per slot, "anywhere"
 *                 when W >= 2N).
 *   SYNTH_HUB     circuit class (the reference's dc1): short rows (1..2K-1
 *                 entries, stratified over the window) + ONE hub row (global
 *                 row N/3) of min(N, max(N/8, 131072)) entries spread over
 *                 all columns + ONE hub column (N/2) present in three rows
 *                 out of four.
 * Values are uniform in [-1, 1); x is uniform in [0, 1).
 */
#ifndef SPMV_SYNTH_H
#define SPMV_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define SYNTH_FN __host__ __device__ static inline
#else
#define SYNTH_FN static inline
#endif

enum synth_kind {
    SYNTH_BANDED = 0,
    SYNTH_RANDOM = 1,
    SYNTH_RAGGED = 2,
    SYNTH_KKT = 3,
    SYNTH_STENCIL = 4,
    SYNTH_POWERLAW = 5,
    SYNTH_HUB = 6,
    SYNTH_KIND_LAST = SYNTH_HUB
};
#define SYNTH_POWERLAW_MAX_K 40 /* (2K+1)^3 << 44 has to fit 64 bits */

typedef struct synth_spec {
    int kind;       /* enum synth_kind */
    int M, N;       /* logical shape of the (shard of the) matrix */
    int K;          /* nominal entries per row */
    int64_t W;      /* column window for the random families */
    int64_t row0;   /* global index of local row 0 (multi-GPU shards) */
    uint64_t seed;  /* matrix seed (42 in BASELINE configs) */
} synth_spec;

/* splitmix64 finaliser over a 64-bit counter */
SYNTH_FN uint64_t synth_mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

SYNTH_FN uint64_t synth_hash2(uint64_t seed, uint64_t a, uint64_t b) {
    return synth_mix(synth_mix(seed ^ (a * 0xD1342543DE82EF95ull)) + b);
}

/* 53-bit uniform in [0,1) */
SYNTH_FN double synth_u01(uint64_t h) {
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

/* x vector element (seed 7 in BASELINE configs), global index */
SYNTH_FN double synth_x(uint64_t seed, int64_t idx) {
    return synth_u01(synth_hash2(seed, 0x78u, (uint64_t)idx));
}

/* grid edge of the stencil family: W if given, else floor(cbrt(N)) */
SYNTH_FN int64_t synth_grid_nx(const synth_spec *s) {
    if (s->W > 0)
        return s->W;
    int64_t n = 1;
    while ((n + 1) * (n + 1) * (n + 1) <= (int64_t)s->N)
        ++n;
    return n;
}

/* neighbours of grid point g inside the grid and below N, ascending; returns
 * the count, writes the columns when cols != 0 */
SYNTH_FN int synth_stencil_cols(const synth_spec *s, int64_t g, int *cols) {
    const int64_t nx = synth_grid_nx(s), nxy = nx * nx;
    const int64_t ix = g % nx, iy = (g / nx) % nx, iz = g / nxy;
    const int full = s->K > 7;
    int n = 0;
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                if (!full && (dx != 0) + (dy != 0) + (dz != 0) > 1)
                    continue;
                const int64_t jx = ix + dx, jy = iy + dy, jz = iz + dz;
                if (jx < 0 || jx >= nx || jy < 0 || jy >= nx || jz < 0)
                    continue;
                const int64_t c = jz * nxy + jy * nx + jx;
                if (c >= (int64_t)s->N)
                    continue;
                if (cols)
                    cols[n] = (int)c;
                ++n;
            }
    return n;
}

/* floor(cbrt(q)), integers only (q < 2^63) */
SYNTH_FN int64_t synth_icbrt(uint64_t q) {
    int64_t r = 0;
    for (int b = 20; b >= 0; --b) {
        const int64_t t = r | ((int64_t)1 << b);
        if ((uint64_t)t * (uint64_t)t * (uint64_t)t <= q)
            r = t;
    }
    return r;
}

/* power-law row length: len = floor(s * u^(-2/3)), u = U / 2^22 with U
 * uniform in 1..2^22, s = (2K+1)/6 -- as the largest len with
 * len^3 <= (2K+1)^3 * 2^44 / (216 * U^2); at least 1, at most N */
SYNTH_FN int synth_powerlaw_len(const synth_spec *s, int64_t g) {
    const uint64_t h = synth_hash2(s->seed, 0x70776cu, (uint64_t)g);
    const uint64_t U = (h >> 42) + 1;
    const uint64_t k = (uint64_t)(2 * (s->K > SYNTH_POWERLAW_MAX_K
                                           ? SYNTH_POWERLAW_MAX_K
                                           : s->K) + 1);
    const uint64_t q = ((k * k * k) << 44) / (216u * U * U);
    int64_t len = synth_icbrt(q);
    if (len < 1)
        len = 1;
    if (len > (int64_t)s->N)
        len = s->N;
    return (int)len;
}

/* the hub family: the hub row / column and what an ordinary row holds */
SYNTH_FN int64_t synth_hub_row(const synth_spec *s) { return (int64_t)s->N / 3; }
SYNTH_FN int synth_hub_col(const synth_spec *s) { return s->N / 2; }
SYNTH_FN int synth_hub_row_len(const synth_spec *s) {
    int64_t len = (int64_t)s->N / 8;
    if (len < 131072)
        len = 131072;
    if (len > (int64_t)s->N)
        len = s->N;
    return (int)len;
}
SYNTH_FN int synth_hub_base_len(const synth_spec *s, int64_t g) {
    const uint64_t h = synth_hash2(s->seed, 0x687562u, (uint64_t)g);
    return 1 + (int)(h % (uint64_t)(2 * s->K - 1));
}
SYNTH_FN int synth_hub_has_col(const synth_spec *s, int64_t g) {
    return (synth_hash2(s->seed, 0x686363u, (uint64_t)g) & 3) != 0;
}

/* number of entries of GLOBAL row g */
SYNTH_FN int synth_row_len(const synth_spec *s, int64_t g) {
    switch (s->kind) {
    case SYNTH_STENCIL:
        return synth_stencil_cols(s, g, 0);
    case SYNTH_POWERLAW:
        return synth_powerlaw_len(s, g);
    case SYNTH_HUB:
        if (g == synth_hub_row(s))
            return synth_hub_row_len(s);
        return synth_hub_base_len(s, g) + synth_hub_has_col(s, g);
    case SYNTH_BANDED:
    case SYNTH_RANDOM:
        return s->K;
    case SYNTH_RAGGED: {
        int q = s->K / 4;
        uint64_t h = synth_hash2(s->seed, 0x6c656eu, (uint64_t)g);
        return s->K - q + (int)(h % (uint64_t)(2 * q + 1));
    }
    default: { /* SYNTH_KKT */
        if ((g & 63) == 17)
            return 8 * s->K;
        int lo = s->K / 4, span = s->K / 4 + 1;
        uint64_t h = synth_hash2(s->seed, 0x6b6b74u, (uint64_t)g);
        return lo + (int)(h % (uint64_t)span);
    }
    }
}

/* value of slot j of GLOBAL row g, uniform in [-1,1) */
SYNTH_FN double synth_val(const synth_spec *s, int64_t g, int j) {
    uint64_t h = synth_hash2(s->seed ^ 0x76616cull, (uint64_t)g, (uint64_t)j);
    return 2.0 * synth_u01(h) - 1.0;
}

/* unsorted column draw t of GLOBAL row g for the random families */
SYNTH_FN int synth_col_draw(const synth_spec *s, int64_t g, int t) {
    int64_t half = s->W / 2;
    int64_t lo = g - half, hi = g + (s->W - half);
    if (lo < 0)
        lo = 0;
    if (hi > s->N)
        hi = s->N;
    if (lo >= hi) { /* window fell off the right edge (M > N shards) */
        lo = 0;
        hi = s->N;
    }
    uint64_t h = synth_hash2(s->seed ^ 0x636f6cull, (uint64_t)g, (uint64_t)t);
    return (int)(lo + (int64_t)(synth_u01(h) * (double)(hi - lo)));
}

/* stratified column: slot j of `len` over [lo, hi), ascending in j */
SYNTH_FN int synth_col_strat(const synth_spec *s, int64_t g, int j, int len,
                             int64_t lo, int64_t hi) {
    const int64_t span = hi - lo;
    const int64_t a = lo + (int64_t)j * span / len;
    const int64_t b = lo + ((int64_t)j + 1) * span / len;
    uint64_t h = synth_hash2(s->seed ^ 0x636f6cull, (uint64_t)g, (uint64_t)j);
    return (int)(a + (int64_t)(synth_u01(h) * (double)(b - a)));
}

/* the window of GLOBAL row g, as synth_col_draw clips it */
SYNTH_FN void synth_window(const synth_spec *s, int64_t g, int64_t *plo,
                           int64_t *phi) {
    int64_t half = s->W / 2;
    int64_t lo = g - half, hi = g + (s->W - half);
    if (lo < 0)
        lo = 0;
    if (hi > s->N)
        hi = s->N;
    if (lo >= hi) {
        lo = 0;
        hi = s->N;
    }
    *plo = lo;
    *phi = hi;
}

/*
 * Fill one row: cols[0..len) ascending, vals[0..len).  `len` must be
 * synth_row_len(s, g).  Insertion sort for the random families: their rows
 * are short (K ~ 16..256); the power-law and hub families, whose rows reach
 * 10^4..10^5 entries, draw ascending columns directly.
 *
 * The _at form places element j at index SYNTH_AT(base + j, skew) of cols /
 * vals, SYNTH_AT(p, k) = p + (p >> k): the device generator stages a range of
 * rows in LDS at their final relative positions, one padding word per 32 so
 * that lanes walking rows of 32 entries do not meet in one bank, and writes
 * the range out with whole-line stores.  skew = 31 is the identity for every
 * index a matrix can have (p < 2^31): the plain form.
 */
#define SYNTH_AT(p, k) ((p) + ((p) >> (k)))
SYNTH_FN void synth_fill_row_at(const synth_spec *s, int64_t g, int len,
                                int *cols, double *vals, int base, int skew) {
#define C_(j) cols[SYNTH_AT(base + (j), skew)]
    if (s->kind == SYNTH_STENCIL) {
        /* neighbours in ascending order: synth_stencil_cols' loop, placed */
        const int64_t nx = synth_grid_nx(s), nxy = nx * nx;
        const int64_t ix = g % nx, iy = (g / nx) % nx, iz = g / nxy;
        const int full = s->K > 7;
        int n = 0;
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    if (!full && (dx != 0) + (dy != 0) + (dz != 0) > 1)
                        continue;
                    const int64_t jx = ix + dx, jy = iy + dy, jz = iz + dz;
                    if (jx < 0 || jx >= nx || jy < 0 || jy >= nx || jz < 0)
                        continue;
                    const int64_t c = jz * nxy + jy * nx + jx;
                    if (c >= (int64_t)s->N)
                        continue;
                    C_(n) = (int)c;
                    ++n;
                }
    } else if (s->kind == SYNTH_POWERLAW) {
        int64_t lo, hi;
        synth_window(s, g, &lo, &hi);
        for (int j = 0; j < len; ++j)
            C_(j) = synth_col_strat(s, g, j, len, lo, hi);
    } else if (s->kind == SYNTH_HUB) {
        if (g == synth_hub_row(s)) {
            for (int j = 0; j < len; ++j)
                C_(j) = synth_col_strat(s, g, j, len, 0, s->N);
        } else {
            const int hub = synth_hub_has_col(s, g);
            const int nbase = len - hub;
            int64_t lo, hi;
            synth_window(s, g, &lo, &hi);
            for (int j = 0; j < nbase; ++j)
                C_(j) = synth_col_strat(s, g, j, nbase, lo, hi);
            if (hub) { /* the hub column goes to its sorted place */
                const int c = synth_hub_col(s);
                int p = nbase;
                while (p > 0 && C_(p - 1) > c) {
                    C_(p) = C_(p - 1);
                    --p;
                }
                C_(p) = c;
            }
        }
    } else if (s->kind == SYNTH_BANDED) {
        int64_t st = g - s->K / 2;
        if (st > (int64_t)s->N - len)
            st = (int64_t)s->N - len;
        if (st < 0)
            st = 0;
        for (int j = 0; j < len; ++j)
            C_(j) = (int)(st + j);
    } else {
        for (int t = 0; t < len; ++t) {
            int c = synth_col_draw(s, g, t);
            int p = t;
            while (p > 0 && C_(p - 1) > c) {
                C_(p) = C_(p - 1);
                --p;
            }
            C_(p) = c;
        }
    }
#undef C_
    for (int j = 0; j < len; ++j)
        vals[SYNTH_AT(base + j, skew)] = synth_val(s, g, j);
}

SYNTH_FN void synth_fill_row(const synth_spec *s, int64_t g, int len,
                             int *cols, double *vals) {
    synth_fill_row_at(s, g, len, cols, vals, 0, 31);
}

#endif /* SPMV_SYNTH_H */

/*
 * gen_kkt_mtx.c -- writes a Matrix Market file with the shape of SuiteSparse
 * Schenk/nlpkkt160 (BASELINE config 4), which cannot be fetched here:
 * a symmetric KKT matrix
 *
 *         [ H  A^T ]      H : n1 x n1   Hessian block
 *     K = [        ]      A : G  x n1   constraint Jacobian
 *         [ A   0  ]
 *
 * of a 3-D PDE-constrained problem on an n x n x n grid: G = n^3 states,
 * B = 6 n^2 boundary controls, n1 = G + B, M = N = n1 + G.  n = 160 gives
 * M = 8 345 600 (nlpkkt160's size), ~1.16e8 stored entries and ~2.27e8
 * entries after the loader mirrors the lower triangle (nlpkkt160: 2.25e8).
 *
 *   H(g,h)      27-point stencil between states; H(c,c) diagonal on controls
 *   A(g,h)      15-point stencil (centre, 6 faces, 8 corners) on states, and
 *               A(g,c) = coupling to each control c whose boundary point is g
 *   value(i,j)  (mix64(i << 32 | j) % 2001 - 1000) / 1000, i >= j: three
 *               decimals, so the text is short and every parser reads the
 *               same double; about 1 entry in 2001 is an explicit zero
 *
 * Like the SuiteSparse files the lower triangle is stored column by column,
 * rows ascending -- the order decides the in-row order of the loaded CSR
 * (reference src/csr.c:91-94,141-145 mirrors while it fills).
 * Rows of the loaded matrix hold 2 (controls), 15-16 (constraints) or up to
 * 42 (states) entries: the irregular-row case of the CSR kernels.
 *
 * The same definition in Python: tests/_kkt.py (row-by-row, for the checks).
 *
 *   gen_kkt_mtx <n> <out.mtx>          write the file
 *   gen_kkt_mtx <n> -                  print "M N entries nnz" only
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static inline int value_k(int64_t i, int64_t j) { /* i >= j; thousandths */
    return (int)(mix64(((uint64_t)i << 32) | (uint64_t)j) % 2001u) - 1000;
}

static char *put_int(char *p, int64_t v) {
    char t[24];
    int n = 0;
    do {
        t[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n)
        *p++ = t[--n];
    return p;
}

static char *put_entry(char *p, int64_t i, int64_t j) { /* 0-based in */
    int k = value_k(i, j);
    p = put_int(p, i + 1);
    *p++ = ' ';
    p = put_int(p, j + 1);
    *p++ = ' ';
    if (k < 0) {
        *p++ = '-';
        k = -k;
    }
    *p++ = (char)('0' + k / 1000);
    *p++ = '.';
    *p++ = (char)('0' + k / 100 % 10);
    *p++ = (char)('0' + k / 10 % 10);
    *p++ = (char)('0' + k % 10);
    *p++ = '\n';
    return p;
}

/* boundary point of control c */
static int64_t control_point(int n, int64_t c) {
    const int64_t f = c / ((int64_t)n * n), r = c % ((int64_t)n * n);
    const int u = (int)(r % n), v = (int)(r / n);
    int x, y, z;
    switch (f) {
    case 0: x = 0; y = u; z = v; break;
    case 1: x = n - 1; y = u; z = v; break;
    case 2: x = u; y = 0; z = v; break;
    case 3: x = u; y = n - 1; z = v; break;
    case 4: x = u; y = v; z = 0; break;
    default: x = u; y = v; z = n - 1; break;
    }
    return x + (int64_t)n * (y + (int64_t)n * z);
}

static int in15(int dx, int dy, int dz) {
    const int s = abs(dx) + abs(dy) + abs(dz);
    return s <= 1 || s == 3;
}

int main(int argc, char **argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: gen_kkt_mtx <n> <out.mtx | ->\n");
        return 2;
    }
    const int n = atoi(argv[1]);
    if (n < 2 || n > 400) {
        fprintf(stderr, "gen_kkt_mtx: n must be in [2, 400]\n");
        return 2;
    }
    const int64_t G = (int64_t)n * n * n, B = 6ll * n * n, n1 = G + B, M = n1 + G;
    /* entries stored (lower triangle) and after mirroring */
    int64_t stored = 0, diag = 0;
    for (int z = 0; z < n; ++z)
        for (int y = 0; y < n; ++y)
            for (int x = 0; x < n; ++x)
                for (int dz = -1; dz <= 1; ++dz)
                    for (int dy = -1; dy <= 1; ++dy)
                        for (int dx = -1; dx <= 1; ++dx) {
                            if (x + dx < 0 || x + dx >= n || y + dy < 0 ||
                                y + dy >= n || z + dz < 0 || z + dz >= n)
                                continue;
                            const int d = dx + n * (dy + n * dz);
                            if (d >= 0)
                                ++stored; /* H, rows >= column */
                            if (in15(dx, dy, dz))
                                ++stored; /* A */
                        }
    diag = G + B;
    stored += B /* H(c,c) */ + B /* A(g,c) */;
    const int64_t nnz = 2 * (stored - diag) + diag;
    if (!strcmp(argv[2], "-")) {
        printf("%lld %lld %lld %lld\n", (long long)M, (long long)M,
               (long long)stored, (long long)nnz);
        return 0;
    }
    FILE *f = fopen(argv[2], "wb");
    if (!f) {
        perror(argv[2]);
        return 1;
    }
    const size_t cap = (size_t)32 << 20;
    char *buf = malloc(cap + 4096);
    if (!buf) {
        fclose(f);
        return 1;
    }
    char *p = buf;
    p += sprintf(p,
                 "%%%%MatrixMarket matrix coordinate real symmetric\n"
                 "%% nlpkkt160-shaped KKT matrix [H A'; A 0], %d^3 grid "
                 "(tools/gen_kkt_mtx.c)\n%lld %lld %lld\n",
                 n, (long long)M, (long long)M, (long long)stored);
    int ok = 1;
#define FLUSH_IF_FULL()                                                       \
    do {                                                                      \
        if ((size_t)(p - buf) >= cap) {                                       \
            ok &= fwrite(buf, 1, (size_t)(p - buf), f) == (size_t)(p - buf);  \
            p = buf;                                                          \
        }                                                                     \
    } while (0)
    /* state columns */
    for (int z = 0; z < n && ok; ++z)
        for (int y = 0; y < n; ++y)
            for (int x = 0; x < n; ++x) {
                const int64_t g = x + (int64_t)n * (y + (int64_t)n * z);
                for (int pass = 0; pass < 2; ++pass) /* H rows, then A rows */
                    for (int dz = -1; dz <= 1; ++dz)
                        for (int dy = -1; dy <= 1; ++dy)
                            for (int dx = -1; dx <= 1; ++dx) {
                                if (x + dx < 0 || x + dx >= n || y + dy < 0 ||
                                    y + dy >= n || z + dz < 0 || z + dz >= n)
                                    continue;
                                const int64_t h = g + dx + (int64_t)n * (dy + (int64_t)n * dz);
                                if (pass == 0) {
                                    if (h >= g)
                                        p = put_entry(p, h, g);
                                } else if (in15(dx, dy, dz)) {
                                    p = put_entry(p, n1 + h, g);
                                }
                            }
                FLUSH_IF_FULL();
            }
    /* control columns */
    for (int64_t c = 0; c < B && ok; ++c) {
        p = put_entry(p, G + c, G + c);
        p = put_entry(p, n1 + control_point(n, c), G + c);
        FLUSH_IF_FULL();
    }
    ok &= fwrite(buf, 1, (size_t)(p - buf), f) == (size_t)(p - buf);
    ok &= fclose(f) == 0;
    free(buf);
    if (!ok) {
        fprintf(stderr, "gen_kkt_mtx: write failed\n");
        return 1;
    }
    printf("%lld %lld %lld %lld\n", (long long)M, (long long)M,
           (long long)stored, (long long)nnz);
    return 0;
}

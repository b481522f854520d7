/*
 * spmv_oracle.c -- CPU restatement of the reference's SpMV path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (spmv_scpa_amd/,
 * include/, the driver) links, loads or calls this file.  It may be used by
 * tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg, and
 * only as the checker.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here bit-for-bit against outputs of the reference's own C code
 * (oracle/_ref/ref_strict, built from /root/reference/src by
 * oracle/build_ref.sh) stored under tests/golden/ (the .ref.txt files).
 *
 * Plain C99 on flat arrays (ctypes-friendly).  Each function names the
 * reference lines it restates.  Strict IEEE: build without -ffast-math.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <errno.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/spmv_synth.h"

#define ORACLE_HACK 32 /* reference include/hll.h:10 */

/* ------------------------------------------------------------------ */
/* Matrix Market header: reference src/mmio.c:93-166 (banner) and      */
/* src/mmio.c:175-200 (size line), as used by src/csr.c:48-59.          */
/* Returns 0 and fills field ('R','P','C','I'), sym ('G','S','K','H'),  */
/* dense flag; non-zero when the banner itself is malformed.            */
/* ------------------------------------------------------------------ */
static void lower(char *s) {
    for (; *s; ++s)
        *s = (char)tolower((unsigned char)*s);
}

static int read_banner(FILE *f, char *field, char *sym, int *dense) {
    char line[1025], tok[5][64];
    if (!fgets(line, sizeof line, f))
        return 1;
    if (sscanf(line, "%63s %63s %63s %63s %63s", tok[0], tok[1], tok[2],
               tok[3], tok[4]) != 5)
        return 1;
    for (int k = 1; k < 5; ++k)
        lower(tok[k]);
    if (strncmp(tok[0], "%%MatrixMarket", 14) != 0)
        return 1;
    if (strcmp(tok[1], "matrix") != 0)
        return 1;
    if (!strcmp(tok[2], "coordinate"))
        *dense = 0;
    else if (!strcmp(tok[2], "array"))
        *dense = 1;
    else
        return 1;
    if (!strcmp(tok[3], "real"))
        *field = 'R';
    else if (!strcmp(tok[3], "complex"))
        *field = 'C';
    else if (!strcmp(tok[3], "pattern"))
        *field = 'P';
    else if (!strcmp(tok[3], "integer"))
        *field = 'I';
    else
        return 1;
    if (!strcmp(tok[4], "general"))
        *sym = 'G';
    else if (!strcmp(tok[4], "symmetric"))
        *sym = 'S';
    else if (!strcmp(tok[4], "hermitian"))
        *sym = 'H';
    else if (!strcmp(tok[4], "skew-symmetric"))
        *sym = 'K';
    else
        return 1;
    return 0;
}

static int read_size(FILE *f, int *M, int *N, int *nz) {
    char line[1025];
    *M = *N = *nz = 0;
    do {
        if (!fgets(line, sizeof line, f))
            return 1;
    } while (line[0] == '%');
    if (sscanf(line, "%d %d %d", M, N, nz) == 3)
        return 0;
    for (;;) {
        int got = fscanf(f, "%d %d %d", M, N, nz);
        if (got == EOF)
            return 1;
        if (got == 3)
            return 0;
    }
}

static void *amalloc(size_t bytes) { /* reference src/utils.c:30-37 */
    void *p = NULL;
    if (posix_memalign(&p, 64, bytes ? bytes : 64))
        return NULL;
    return p;
}

void oracle_free(void *p) { free(p); }

/* reference src/csr.c:18-30 */
void oracle_matrix_name(const char *path, char out[64]) {
    const char *base = strrchr(path, '/');
    base = base ? base + 1 : path;
    size_t n = strlen(base);
    if (n > 4 && !strcmp(base + n - 4, ".mtx"))
        n -= 4;
    if (n > 63)
        n = 63;
    memcpy(out, base, n);
    out[n] = 0;
}

/*
 * reference src/csr.c:31-171: two passes over the text; entries are
 * scattered into their rows in FILE order; a symmetric file mirrors each
 * off-diagonal entry right after the entry itself; pattern entries get 1.0;
 * 1-based -> 0-based; out-of-range -> -ERANGE; short/garbled -> -EIO;
 * anything but "matrix coordinate real|pattern" -> -EINVAL.
 * Returns 0 or a negative errno; arrays are malloc'ed (oracle_free).
 */
int oracle_load_mtx(const char *path, int *M_, int *N_, int *NZ_, int **IRP_,
                    int **JA_, double **AS_) {
    FILE *f = fopen(path, "r");
    if (!f)
        return -errno;
    char field = 0, sym = 0;
    int dense = 0, M, N, nz0, rc = 0;
    int *cnt = NULL, *IRP = NULL, *JA = NULL;
    double *AS = NULL;
    if (read_banner(f, &field, &sym, &dense) || dense ||
        !(field == 'R' || field == 'P')) {
        rc = -EINVAL;
        goto out;
    }
    if (read_size(f, &M, &N, &nz0)) {
        rc = -EINVAL;
        goto out;
    }
    int mirror = (sym == 'S'), pat = (field == 'P');
    long data_pos = ftell(f);
    cnt = calloc((size_t)M > 0 ? (size_t)M : 1, sizeof *cnt);
    if (!cnt) {
        rc = -ENOMEM;
        goto out;
    }
    long total = 0;
    for (int pass = 0; pass < 2 && !rc; ++pass) {
        if (pass == 1) {
            IRP = amalloc(((size_t)M + 1) * sizeof(int));
            JA = amalloc((size_t)total * sizeof(int));
            AS = amalloc((size_t)total * sizeof(double));
            if (!IRP || !JA || !AS) {
                rc = -ENOMEM;
                break;
            }
            IRP[0] = 0;
            for (int r = 0; r < M; ++r)
                IRP[r + 1] = IRP[r] + cnt[r];
            memset(cnt, 0, (size_t)M * sizeof *cnt);
            if (fseek(f, data_pos, SEEK_SET)) {
                rc = -EIO;
                break;
            }
        }
        for (int e = 0; e < nz0; ++e) {
            int i, j;
            double v = 1.0;
            int ok = pat ? fscanf(f, "%d %d", &i, &j) == 2
                         : fscanf(f, "%d %d %lf", &i, &j, &v) == 3;
            if (!ok) {
                rc = -EIO;
                break;
            }
            --i;
            --j;
            if (pass == 0) {
                if (i < 0 || i >= M || j < 0 || j >= N) {
                    rc = -ERANGE;
                    break;
                }
                cnt[i]++;
                total++;
                if (mirror && i != j) {
                    cnt[j]++;
                    total++;
                }
            } else {
                long p = (long)IRP[i] + cnt[i]++;
                JA[p] = j;
                AS[p] = v;
                if (mirror && i != j) {
                    long q = (long)IRP[j] + cnt[j]++;
                    JA[q] = i;
                    AS[q] = v;
                }
            }
        }
    }
out:
    free(cnt);
    fclose(f);
    if (rc) {
        free(IRP);
        free(JA);
        free(AS);
        return rc;
    }
    *M_ = M;
    *N_ = N;
    *NZ_ = (int)total;
    *IRP_ = IRP;
    *JA_ = JA;
    *AS_ = AS;
    return 0;
}

/* reference src/csr.c:201-216 -- THE parity oracle: left-to-right, stored
 * order, one multiply and one add per entry. */
void oracle_csr_spmv(int M, const int *IRP, const int *JA, const double *AS,
                     const double *x, double *y) {
    for (int i = 0; i < M; ++i) {
        double acc = 0.0;
        for (int k = IRP[i]; k < IRP[i + 1]; ++k)
            acc += AS[k] * x[JA[k]];
        y[i] = acc;
    }
}

/* Row scale sum_j |a_ij x_j| used by the parity metric. */
void oracle_csr_abs_spmv(int M, const int *IRP, const int *JA,
                         const double *AS, const double *x, double *s) {
    for (int i = 0; i < M; ++i) {
        double acc = 0.0;
        for (int k = IRP[i]; k < IRP[i + 1]; ++k)
            acc += fabs(AS[k] * x[JA[k]]);
        s[i] = acc;
    }
}

/* reference src/csr.c:278-298 (rows over threads; same per-row order) */
void oracle_csr_spmv_omp(int M, const int *IRP, const int *JA,
                         const double *AS, const double *x, double *y,
                         int threads) {
    (void)threads;
#pragma omp parallel for schedule(guided) num_threads(threads)
    for (int i = 0; i < M; ++i) {
        double acc = 0.0;
        for (int k = IRP[i]; k < IRP[i + 1]; ++k)
            acc += AS[k] * x[JA[k]];
        y[i] = acc;
    }
}

/*
 * reference src/csr.c:218-276: greedy nnz-balanced row cut.  starts has
 * room for *threads+1 ints; on return *threads may have shrunk.
 */
void oracle_partition_rows(int M, const int *IRP, int *threads, int *starts) {
    int max_t = *threads, t = 0;
    long total = (long)IRP[M] - IRP[0];
    double target = (double)total / max_t, running = 0.0;
    starts[0] = 0;
    for (int r = 0; r < M && t < max_t - 1; ++r) {
        running += IRP[r + 1] - IRP[r];
        if (running >= target) {
            starts[++t] = r + 1;
            running = 0.0;
        }
    }
    starts[t + 1] = M;
    *threads = t + 1;
}

/* ------------------------------------------------------------------ */
/* HLL: reference src/hll.c:19-95.  Flat form: block b owns slots       */
/* [off[b], off[b+1]) with rows_b = min(32, M-32b) rows and maxnz[b]     */
/* columns; pad JA = -1, AS = 0.0; row-major idx = i*maxnz+j, col-major  */
/* idx = j*rows_b+i (rows_b is the ACTUAL row count of the block).       */
/* ------------------------------------------------------------------ */
int oracle_hll_num_blocks(int M) { return (M + ORACLE_HACK - 1) / ORACLE_HACK; }

/* fills off[nb+1], maxnz[nb], blknz[nb]; returns total slots */
int64_t oracle_hll_layout(int M, const int *IRP, int64_t *off, int *maxnz,
                          int *blknz) {
    int nb = oracle_hll_num_blocks(M);
    int64_t s = 0;
    for (int b = 0; b < nb; ++b) {
        int r0 = b * ORACLE_HACK;
        int r1 = r0 + ORACLE_HACK < M ? r0 + ORACLE_HACK : M;
        int mx = 0, tot = 0;
        for (int i = r0; i < r1; ++i) {
            int len = IRP[i + 1] - IRP[i];
            tot += len;
            if (len > mx)
                mx = len;
        }
        off[b] = s;
        maxnz[b] = mx;
        blknz[b] = tot;
        s += (int64_t)(r1 - r0) * mx;
    }
    off[nb] = s;
    return s;
}

void oracle_csr_to_hll(int M, const int *IRP, const int *JA, const double *AS,
                       int col_major, const int64_t *off, const int *maxnz,
                       int *HJA, double *HAS) {
    int nb = oracle_hll_num_blocks(M);
    for (int b = 0; b < nb; ++b) {
        int r0 = b * ORACLE_HACK;
        int rows = (r0 + ORACLE_HACK < M ? r0 + ORACLE_HACK : M) - r0;
        int mx = maxnz[b];
        int *bj = HJA + off[b];
        double *ba = HAS + off[b];
        for (int64_t t = 0; t < (int64_t)rows * mx; ++t) {
            bj[t] = -1;
            ba[t] = 0.0;
        }
        for (int i = 0; i < rows; ++i) {
            int st = IRP[r0 + i], len = IRP[r0 + i + 1] - st;
            for (int j = 0; j < len; ++j) {
                int64_t t = col_major ? (int64_t)j * rows + i
                                      : (int64_t)i * mx + j;
                bj[t] = JA[st + j];
                ba[t] = AS[st + j];
            }
        }
    }
}

/* reference src/hll.c:127-150 (row-major) and 152-176 (col-major): pads
 * (column -1) are skipped, accumulation left to right. */
void oracle_hll_spmv(int M, int col_major, const int64_t *off,
                     const int *maxnz, const int *HJA, const double *HAS,
                     const double *x, double *y) {
    int nb = oracle_hll_num_blocks(M);
    for (int b = 0; b < nb; ++b) {
        int r0 = b * ORACLE_HACK;
        int rows = (r0 + ORACLE_HACK < M ? r0 + ORACLE_HACK : M) - r0;
        int mx = maxnz[b];
        const int *bj = HJA + off[b];
        const double *ba = HAS + off[b];
        for (int i = 0; i < rows; ++i) {
            double acc = 0.0;
            for (int j = 0; j < mx; ++j) {
                int64_t t = col_major ? (int64_t)j * rows + i
                                      : (int64_t)i * mx + j;
                int c = bj[t];
                if (c != -1)
                    acc += ba[t] * x[c];
            }
            y[r0 + i] = acc;
        }
    }
}

/*
 * reference src/cuda_hll.cu:173-195: what the reference's GPU upload does
 * to pad slots -- a pad takes the previous valid column of its row, or 0
 * for a row with no entries -- so device kernels can run branch-free.
 */
void oracle_hll_fix_pads(int M, int col_major, const int64_t *off,
                         const int *maxnz, int *HJA) {
    int nb = oracle_hll_num_blocks(M);
    for (int b = 0; b < nb; ++b) {
        int r0 = b * ORACLE_HACK;
        int rows = (r0 + ORACLE_HACK < M ? r0 + ORACLE_HACK : M) - r0;
        int mx = maxnz[b];
        int *bj = HJA + off[b];
        for (int i = 0; i < rows; ++i) {
            int last = 0;
            for (int j = 0; j < mx; ++j) {
                int64_t t = col_major ? (int64_t)j * rows + i
                                      : (int64_t)i * mx + j;
                if (bj[t] == -1)
                    bj[t] = last;
                else
                    last = bj[t];
            }
        }
    }
}

/* reference src/vector.c:36-41 with main.c:97-102: x = rand()/RAND_MAX from
 * the never-seeded glibc generator (== srand(1)). */
void oracle_rand_x(double *x, size_t n) {
    srand(1);
    for (size_t i = 0; i < n; ++i)
        x[i] = (double)rand() / RAND_MAX;
}

/* reference include/utils.h:70-75 */
double oracle_gflops(double ms, int nnz) {
    return ms <= 0.0 ? 0.0 : (2.0 * nnz) / (ms * 1e6);
}

/* reference src/utils.c:39-60: 0 when ||a-b||_2 <= 0.1, else -1 */
int oracle_validate(const double *a, size_t na, const double *b, size_t nb) {
    if (na != nb)
        return -1;
    double s = 0.0;
    for (size_t i = 0; i < na; ++i)
        s += (a[i] - b[i]) * (a[i] - b[i]);
    return sqrt(s) > 1e-1 ? -1 : 0;
}

/* ------------------------------------------------------------------ */
/* Synthetic workloads (include/spmv_synth.h) into flat CSR arrays.     */
/* ------------------------------------------------------------------ */
int64_t oracle_synth_nnz(int kind, int M, int N, int K, int64_t W,
                         int64_t row0, uint64_t seed) {
    synth_spec s = {kind, M, N, K, W, row0, seed};
    int64_t nz = 0;
    for (int i = 0; i < M; ++i)
        nz += synth_row_len(&s, row0 + i);
    return nz;
}

void oracle_synth_csr(int kind, int M, int N, int K, int64_t W, int64_t row0,
                      uint64_t seed, int *IRP, int *JA, double *AS) {
    synth_spec s = {kind, M, N, K, W, row0, seed};
    IRP[0] = 0;
    for (int i = 0; i < M; ++i)
        IRP[i + 1] = IRP[i] + synth_row_len(&s, row0 + i);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < M; ++i)
        synth_fill_row(&s, row0 + i, IRP[i + 1] - IRP[i], JA + IRP[i],
                       AS + IRP[i]);
}

void oracle_synth_x(uint64_t seed, int64_t first, int64_t n, double *x) {
    for (int64_t i = 0; i < n; ++i)
        x[i] = synth_x(seed, first + i);
}

/* One synthetic row's exact serial dot product: lets tests check rows of a
 * 10M-row device result without materialising the matrix on the host. */
double oracle_synth_row_dot(int kind, int M, int N, int K, int64_t W,
                            int64_t row0, uint64_t seed, uint64_t xseed,
                            int64_t grow, double *abs_out) {
    synth_spec s = {kind, M, N, K, W, row0, seed};
    int len = synth_row_len(&s, grow);
    int *c = malloc((size_t)(len > 0 ? len : 1) * sizeof *c);
    double *v = malloc((size_t)(len > 0 ? len : 1) * sizeof *v);
    synth_fill_row(&s, grow, len, c, v);
    double acc = 0.0, sab = 0.0;
    for (int j = 0; j < len; ++j) {
        double p = v[j] * synth_x(xseed, c[j]);
        acc += p;
        sab += fabs(p);
    }
    free(c);
    free(v);
    if (abs_out)
        *abs_out = sab;
    return acc;
}

/* Wall-clock timing of the port, for bench.py's cpu_baseline (kind "port"). */
double oracle_time_csr_ms(int M, const int *IRP, const int *JA,
                          const double *AS, const double *x, double *y,
                          int threads, int reps) {
    struct timespec a, b;
    double best = 1e300;
    for (int r = 0; r < reps; ++r) {
        clock_gettime(CLOCK_MONOTONIC, &a);
        if (threads <= 1)
            oracle_csr_spmv(M, IRP, JA, AS, x, y);
        else
            oracle_csr_spmv_omp(M, IRP, JA, AS, x, y, threads);
        clock_gettime(CLOCK_MONOTONIC, &b);
        double ms = (b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6;
        if (ms < best)
            best = ms;
    }
    return best;
}

/*
 * csr.c -- CSR container, Matrix Market loader, synthetic generator, row
 * partitions and the CPU / GPU benchmark wrappers (API: include/csr.h).
 *
 * Behavioural reference: src/csr.c of 0xmenna/spmv-scpa (cited per
 * function).  The implementation is new: one pass over an in-memory image
 * of the file and a stable counting sort by row, instead of two fscanf
 * passes over the text.
 */
#include <errno.h>
#include <limits.h>
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "csr.h"
#include "err.h"
#include "hip_csr.h"
#include "mm_header.h"
#include "spmv_engine.h"
#include "spmv_synth.h"

/* ------------------------------------------------------------------ */
/* container                                                            */
/* ------------------------------------------------------------------ */

void extract_matrix_name(const char *path, char *name_out) {
    const char *slash = strrchr(path, '/');
    const char *base = slash ? slash + 1 : path;
    size_t n = strlen(base);
    if (n > 4 && memcmp(base + n - 4, ".mtx", 4) == 0)
        n -= 4;
    if (n > MAX_NAME - 1)
        n = MAX_NAME - 1;
    memcpy(name_out, base, n);
    name_out[n] = '\0';
}

sparse_csr *csr_alloc(const char *name, int M, int N, int NZ) {
    if (M < 0 || N < 0 || NZ < 0)
        return ERR_PTR(-EINVAL);
    sparse_csr *A = malloc(sizeof *A);
    int *irp = aligned_malloc(((size_t)M + 1) * sizeof(int));
    int *ja = aligned_malloc((size_t)NZ * sizeof(int));
    double *as = aligned_malloc((size_t)NZ * sizeof(double));
    if (!A || !irp || !ja || !as) {
        free(A);
        free(irp);
        free(ja);
        free(as);
        return ERR_PTR(-ENOMEM);
    }
    memset(irp, 0, ((size_t)M + 1) * sizeof(int));
    init_csr(A, name ? name : "", M, N, NZ, irp, ja, as);
    return A;
}

void csr_free(sparse_csr *A) {
    if (IS_ERR_OR_NULL(A))
        return;
    free(A->IRP);
    free(A->JA);
    free(A->AS);
    free(A);
}

/* ------------------------------------------------------------------ */
/* Matrix Market loader                                                 */
/* ------------------------------------------------------------------ */

static int is_space(int c) {
    return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' ||
           c == '\f';
}

/* "%d": optional blanks, optional sign, at least one digit */
static int scan_int(const char **pp, const char *end, int *out) {
    const char *p = *pp;
    while (p < end && is_space((unsigned char)*p))
        ++p;
    int neg = 0;
    if (p < end && (*p == '-' || *p == '+'))
        neg = (*p++ == '-');
    if (p >= end || *p < '0' || *p > '9')
        return 0;
    long long v = 0;
    while (p < end && *p >= '0' && *p <= '9') {
        v = v * 10 + (*p++ - '0');
        if (v > (long long)INT_MAX + 1)
            v = (long long)INT_MAX + 1; /* saturate; caught by range check */
    }
    if (neg)
        v = -v;
    if (v > INT_MAX)
        v = INT_MAX;
    if (v < INT_MIN)
        v = INT_MIN;
    *out = (int)v;
    *pp = p;
    return 1;
}

/* "%lf": strtod accepts what scanf accepts (decimal, exponent, hex, inf,
 * nan); the buffer is NUL-terminated so it cannot run past `end`. */
static int scan_double(const char **pp, const char *end, double *out) {
    const char *p = *pp;
    while (p < end && is_space((unsigned char)*p))
        ++p;
    if (p >= end)
        return 0;
    char *stop;
    double v = strtod(p, &stop);
    if (stop == p)
        return 0;
    *out = v;
    *pp = stop;
    return 1;
}

/*
 * Correctly rounded conversion of a plain decimal token "[+-]ddd[.ddd]"
 * without strtod: with at most 15 significant digits the digits form an
 * integer below 2^53 and 10^k (k <= 22 fraction digits) is exact in binary64,
 * so ONE IEEE division rounds once -- the value strtod returns (Clinger's
 * fast path).  Returns 0 (caller uses strtod) for anything else: exponents,
 * hex, inf/nan, longer mantissas.
 */
static int fast_decimal(const char *p, const char *e, double *out) {
    static const double p10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,
                                   1e8,  1e9,  1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                                   1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    int neg = 0, digits = 0, frac = 0, seen_dot = 0, any = 0;
    unsigned long long m = 0;
    if (p < e && (*p == '-' || *p == '+'))
        neg = (*p++ == '-');
    for (; p < e; ++p) {
        const char c = *p;
        if (c >= '0' && c <= '9') {
            any = 1;
            if (m || c != '0')
                ++digits; /* significant digits: leading zeros are free */
            if (digits > 15)
                return 0;
            m = m * 10 + (unsigned)(c - '0');
            frac += seen_dot;
        } else if (c == '.' && !seen_dot) {
            seen_dot = 1;
        } else {
            return 0;
        }
    }
    if (!any || frac > 22)
        return 0;
    const double v = (double)m / p10[frac];
    *out = neg ? -v : v;
    return 1;
}

/*
 * The file image.  Regular files are mapped (no copy; the pages are faulted
 * in by the threads that tokenise them); the parsers need a NUL behind the
 * text, which the zero-filled tail of the last page provides unless the size
 * is an exact multiple of the page size -- then, and for pipes, the file is
 * read into a buffer.
 */
struct file_image {
    char *text;
    size_t len;
    size_t mapped; /* bytes to munmap; 0: `text` is malloc'ed */
};

static void image_release(struct file_image *im) {
    if (im->mapped)
        munmap(im->text, im->mapped);
    else
        free(im->text);
    im->text = NULL;
}

static int read_whole_file(const char *path, char **buf, size_t *len);

static int image_open(const char *path, struct file_image *im) {
    memset(im, 0, sizeof *im);
    int fd = open(path, O_RDONLY);
    if (fd < 0)
        return -errno;
    struct stat st;
    const long page = sysconf(_SC_PAGESIZE);
    /* A mapped file that another process truncates or rewrites while the
     * tokenisers run ends the loader with SIGBUS instead of an error code.
     * Writers are required to write under a temporary name and rename
     * (include/csr.h; tools/gen_kkt_mtx.c callers and the sidecar writer do);
     * as a second line of defence a file modified within the last two
     * seconds -- possibly still being written -- is READ, not mapped: a
     * short read is an ordinary -EIO / parse error. */
    struct timespec now;
    clock_gettime(CLOCK_REALTIME, &now);
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0 &&
        page > 0 && (st.st_size % page) != 0 &&
        now.tv_sec - st.st_mtim.tv_sec > 2) {
        void *m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) {
            close(fd);
            (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
            im->text = m;
            im->len = (size_t)st.st_size;
            im->mapped = (size_t)st.st_size;
            return 0;
        }
    }
    close(fd);
    return read_whole_file(path, &im->text, &im->len);
}

static int read_whole_file(const char *path, char **buf, size_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f)
        return -errno;
    size_t cap = 1 << 16, n = 0;
    struct stat st; /* size the buffer once when the size is known */
    if (fstat(fileno(f), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0)
        cap = (size_t)st.st_size + 1;
    char *b = malloc(cap + 1);
    if (!b) {
        fclose(f);
        return -ENOMEM;
    }
    for (;;) {
        size_t got = fread(b + n, 1, cap - n, f);
        n += got;
        if (got == 0)
            break;
        if (n == cap) {
            cap *= 2;
            char *nb = realloc(b, cap + 1);
            if (!nb) {
                free(b);
                fclose(f);
                return -ENOMEM;
            }
            b = nb;
        }
    }
    int bad = ferror(f);
    fclose(f);
    if (bad) {
        free(b);
        return -EIO;
    }
    b[n] = '\0';
    *buf = b;
    *len = n;
    return 0;
}

/*
 * Parallel parse of the entry list: the text is cut into one piece per
 * thread at token boundaries, tokens are counted, then every thread converts
 * its own tokens (token g belongs to entry g / fields, field g % fields).
 * Only "clean" files take this path -- every token is consumed entirely by
 * its field's parser.  Anything else (a token like "1-2" that fscanf would
 * split, garbage, ...) makes the function return 0 and the caller runs the
 * sequential scanner, whose behaviour is the reference's by construction.
 * Returns 1 when it parsed `*complete` whole entries (<= nz0).
 */
static int parse_entries_parallel(const char *text, size_t off, size_t len,
                                  int nz0, int pattern, int *ei, int *ej,
                                  double *ev, int *complete) {
    int T = spmv_host_threads();
    if (T > 64)
        T = 64;
    if (T < 2 || nz0 < 100000)
        return 0;
    const int fields = pattern ? 2 : 3;
    size_t start[65];
    long long ntok[65], first[66];
    for (int t = 0; t <= T; ++t) {
        size_t s = off + (len - off) / (size_t)T * (size_t)t;
        if (t == T)
            s = len;
        else if (t > 0) /* a token belongs to the piece it starts in */
            while (s < len && !is_space((unsigned char)text[s]) &&
                   !is_space((unsigned char)text[s - 1]))
                ++s;
        start[t] = s;
    }
#pragma omp parallel for num_threads(T) schedule(static, 1)
    for (int t = 0; t < T; ++t) {
        long long n = 0;
        int in = 0;
        for (size_t k = start[t]; k < start[t + 1]; ++k) {
            int sp = is_space((unsigned char)text[k]);
            n += (!sp && !in);
            in = !sp;
        }
        ntok[t] = n;
    }
    first[0] = 0;
    for (int t = 0; t < T; ++t)
        first[t + 1] = first[t] + ntok[t];
    long long whole = first[T] / fields;
    if (whole > nz0)
        whole = nz0;
    const long long need = whole * fields;
    int dirty = 0;
#pragma omp parallel for num_threads(T) schedule(static, 1) reduction(| : dirty)
    for (int t = 0; t < T; ++t) {
        long long g = first[t];
        size_t k = start[t];
        const size_t kend = start[t + 1];
        while (k < kend && g < need && !dirty) {
            while (k < kend && is_space((unsigned char)text[k]))
                ++k;
            if (k >= kend)
                break;
            size_t e = k;
            while (e < len && !is_space((unsigned char)text[e]))
                ++e;
            const long long ent = g / fields;
            const int f = (int)(g % fields);
            const char *p = text + k;
            if (f < 2) {
                int v;
                if (!scan_int(&p, text + e, &v) || p != text + e)
                    dirty = 1;
                else if (f == 0)
                    ei[ent] = v;
                else
                    ej[ent] = v;
            } else {
                char *stop;
                double v;
                if (fast_decimal(p, text + e, &v)) {
                    ev[ent] = v;
                } else {
                    v = strtod(p, &stop);
                    if (stop != text + e)
                        dirty = 1;
                    else
                        ev[ent] = v;
                }
            }
            ++g;
            k = e;
        }
    }
    if (dirty)
        return 0;
    *complete = (int)whole;
    return 1;
}

sparse_csr *io_load_csr(const char *path) {
    struct file_image im;
    int rc = image_open(path, &im);
    if (rc)
        return ERR_PTR(rc);
    const char *text = im.text;
    const size_t len = im.len;

    int *ei = NULL, *ej = NULL, *fill = NULL;
    double *ev = NULL;
    sparse_csr *A = NULL;
    mm_info mm;

    /* reference csr.c:48-57: only "matrix coordinate real|pattern" */
    if (mm_parse_header(text, len, &mm) != MM_OK || mm.object != 'M' ||
        mm.format != 'C' || !(mm.field == 'R' || mm.field == 'P') ||
        mm.rows < 0 || mm.cols < 0 || mm.entries < 0) {
        rc = -EINVAL;
        goto done;
    }
    const int M = mm.rows, N = mm.cols, nz0 = mm.entries;
    const int mirror = (mm.symmetry == 'S'); /* csr.c:58: 'K','H' as general */
    const int pattern = (mm.field == 'P');

    ei = malloc(((size_t)nz0 + 1) * sizeof *ei);
    ej = malloc(((size_t)nz0 + 1) * sizeof *ej);
    ev = pattern ? NULL : malloc(((size_t)nz0 + 1) * sizeof *ev);
    fill = calloc((size_t)M + 1, sizeof *fill);
    if (!ei || !ej || (!pattern && !ev) || !fill) {
        rc = -ENOMEM;
        goto done;
    }

    /* parse (file order kept).  Large clean files are tokenised in parallel;
     * everything else goes through the sequential scanner. */
    int parsed = 0; /* whole entries converted so far */
    int fast = parse_entries_parallel(text, mm.data_offset, len, nz0, pattern,
                                      ei, ej, ev, &parsed);
    if (!fast) {
        const char *p = text + mm.data_offset, *end = text + len;
        for (parsed = 0; parsed < nz0; ++parsed) {
            int i, j;
            double v = 1.0;
            if (!scan_int(&p, end, &i) || !scan_int(&p, end, &j) ||
                (!pattern && !scan_double(&p, end, &v)))
                break; /* -EIO below, unless an earlier entry is out of range */
            ei[parsed] = i;
            ej[parsed] = j;
            if (!pattern)
                ev[parsed] = v;
        }
    }
    /* validate and count per row, in entry order: the first bad entry
     * decides the code, as in the reference's first pass (csr.c:68-95) */
    long long stored = 0;
    for (int e = 0; e < parsed; ++e) {
        const int i = --ei[e], j = --ej[e];
        if (i < 0 || i >= M || j < 0 || j >= N) {
            rc = -ERANGE; /* csr.c:84-87 */
            goto done;
        }
        fill[i]++;
        stored++;
        if (mirror && i != j) {
            fill[j]++;
            stored++;
        }
    }
    if (parsed < nz0) {
        rc = -EIO; /* csr.c:71-79: short or unparsable entry */
        goto done;
    }
    if (stored > INT_MAX) {
        rc = -EOVERFLOW;
        goto done;
    }

    A = csr_alloc("", M, N, (int)stored);
    if (IS_ERR(A)) {
        rc = PTR_ERR(A);
        A = NULL;
        goto done;
    }
    extract_matrix_name(path, A->name);
    for (int r = 0; r < M; ++r)
        A->IRP[r + 1] = A->IRP[r] + fill[r];
    memset(fill, 0, ((size_t)M + 1) * sizeof *fill);

    /* stable scatter: an entry, then its mirror image (csr.c:138-145).
     * Every thread walks the whole entry list in file order and places what
     * lands in ITS row range (ranges of about equal nnz): the order inside a
     * row is the sequential one, rows have one writer, no atomics -- and the
     * fresh pages of JA/AS are first touched by the thread that fills them. */
    {
        int T = nz0 >= 100000 ? spmv_host_threads() : 1;
        if (T > 64)
            T = 64;
#pragma omp parallel num_threads(T)
        {
            const int t = omp_get_thread_num(), nt = omp_get_num_threads();
            int r0, r1;
            if (nt == 1) {
                r0 = 0;
                r1 = M;
            } else { /* first row whose offset reaches the thread's share */
                int cut[2];
                for (int s = 0; s < 2; ++s) {
                    const long long want = stored * (long long)(t + s) / nt;
                    int lo = 0, hi = M;
                    while (lo < hi) {
                        int mid = lo + (hi - lo) / 2;
                        if (A->IRP[mid] < want)
                            lo = mid + 1;
                        else
                            hi = mid;
                    }
                    cut[s] = (t + s == nt) ? M : lo;
                }
                r0 = cut[0];
                r1 = cut[1];
            }
            for (int e = 0; e < nz0; ++e) {
                const int i = ei[e], j = ej[e];
                if (i >= r0 && i < r1) {
                    const size_t at = (size_t)A->IRP[i] + (size_t)fill[i]++;
                    A->JA[at] = j;
                    A->AS[at] = pattern ? 1.0 : ev[e];
                }
                if (mirror && i != j && j >= r0 && j < r1) {
                    const size_t bt = (size_t)A->IRP[j] + (size_t)fill[j]++;
                    A->JA[bt] = i;
                    A->AS[bt] = pattern ? 1.0 : ev[e];
                }
            }
        }
    }

done:
    image_release(&im);
    free(ei);
    free(ej);
    free(ev);
    free(fill);
    if (rc) {
        csr_free(A);
        return ERR_PTR(rc);
    }
    return A;
}

/* ------------------------------------------------------------------ */
/* binary sidecar: the three CSR arrays behind a small header           */
/* ------------------------------------------------------------------ */

#define CSR_BIN_MAGIC 0x32525343564d5053ull /* "SPMVCSR2" */

/*
 * The sidecar is a cache of a parse, never a second source of truth: it
 * names the text file it was made from (size and mtime to the nanosecond,
 * exact match required), it is validated structurally before anything is
 * handed on (a corrupt or foreign file must not reach a GPU kernel, where
 * x[JA[k]] would fault), and it appears atomically (written under a
 * temporary name, then rename()d: ranks that load the same .mtx at the same
 * time each write their own temporary and the last rename wins whole).
 */
struct csr_bin_header {
    uint64_t magic;
    int32_t M, N, NZ, pad;
    int64_t src_size;       /* bytes of the .mtx; -1: not tied to a file */
    int64_t src_mtime_sec;  /* st_mtim of the .mtx */
    int64_t src_mtime_nsec;
    char name[MAX_NAME];
};

static int csr_write_bin(const sparse_csr *A, const char *path,
                         const struct stat *src) {
    if (IS_ERR_OR_NULL(A) || !path)
        return -EINVAL;
    char tmp[MAX_PATH * 4 + 32];
    if (snprintf(tmp, sizeof tmp, "%s.tmp.%ld", path, (long)getpid()) >=
        (int)sizeof tmp)
        return -ENAMETOOLONG;
    FILE *f = fopen(tmp, "wb");
    if (!f)
        return -errno;
    struct csr_bin_header h;
    memset(&h, 0, sizeof h);
    h.magic = CSR_BIN_MAGIC;
    h.M = A->M;
    h.N = A->N;
    h.NZ = A->NZ;
    h.src_size = src ? (int64_t)src->st_size : -1;
    h.src_mtime_sec = src ? (int64_t)src->st_mtim.tv_sec : 0;
    h.src_mtime_nsec = src ? (int64_t)src->st_mtim.tv_nsec : 0;
    memcpy(h.name, A->name, MAX_NAME);
    /* the error code is taken AT the failing call (errno of an earlier,
     * unrelated failure must not leak into the result) */
    int err = 0;
    errno = 0;
    if (fwrite(&h, sizeof h, 1, f) != 1 ||
        fwrite(A->IRP, sizeof(int), (size_t)A->M + 1, f) != (size_t)A->M + 1 ||
        fwrite(A->JA, sizeof(int), (size_t)A->NZ, f) != (size_t)A->NZ ||
        fwrite(A->AS, sizeof(double), (size_t)A->NZ, f) != (size_t)A->NZ)
        err = errno ? -errno : -EIO;
    errno = 0;
    if (fclose(f) != 0 && !err)
        err = errno ? -errno : -EIO;
    errno = 0;
    if (!err && rename(tmp, path) != 0)
        err = errno ? -errno : -EIO;
    if (err)
        (void)remove(tmp);
    return err;
}

int csr_save_bin(const sparse_csr *A, const char *path) {
    return csr_write_bin(A, path, NULL);
}

/* IRP starts at 0, never decreases, ends at NZ; every column is in [0, N) */
static int csr_arrays_valid(const sparse_csr *A) {
    const int M = A->M, N = A->N, NZ = A->NZ;
    if (A->IRP[0] != 0 || A->IRP[M] != NZ)
        return 0;
    int bad = 0;
#pragma omp parallel for reduction(| : bad) schedule(static) num_threads(spmv_host_threads())
    for (int i = 0; i < M; ++i)
        bad |= A->IRP[i + 1] < A->IRP[i];
#pragma omp parallel for reduction(| : bad) schedule(static) num_threads(spmv_host_threads())
    for (int k = 0; k < NZ; ++k)
        bad |= (unsigned)A->JA[k] >= (unsigned)N;
    return !bad;
}

static sparse_csr *csr_read_bin(const char *path, const struct stat *src) {
    FILE *f = fopen(path, "rb");
    if (!f)
        return ERR_PTR(-errno);
    struct csr_bin_header h;
    struct stat st;
    sparse_csr *A = NULL;
    int rc = 0;
    if (fread(&h, sizeof h, 1, f) != 1 || h.magic != CSR_BIN_MAGIC ||
        h.M < 0 || h.N < 0 || h.NZ < 0) {
        rc = -EINVAL;
        goto out;
    }
    /* made from exactly this text file? (size and mtime to the nanosecond) */
    if (src && (h.src_size != (int64_t)src->st_size ||
                h.src_mtime_sec != (int64_t)src->st_mtim.tv_sec ||
                h.src_mtime_nsec != (int64_t)src->st_mtim.tv_nsec)) {
        rc = -ESTALE;
        goto out;
    }
    /* the file must hold exactly the arrays the header promises */
    if (fstat(fileno(f), &st) != 0 ||
        (uint64_t)st.st_size !=
            sizeof h + ((uint64_t)h.M + 1) * sizeof(int) +
                (uint64_t)h.NZ * (sizeof(int) + sizeof(double))) {
        rc = -EIO;
        goto out;
    }
    h.name[MAX_NAME - 1] = '\0';
    A = csr_alloc(h.name, h.M, h.N, h.NZ);
    if (IS_ERR(A)) {
        rc = PTR_ERR(A);
        A = NULL;
        goto out;
    }
    {
        /* the three arrays lie back to back behind the header; pread them in
         * 32 MiB pieces from all threads (first touch + copy in parallel:
         * 2.8 GB of nlpkkt160-sized arrays in 0.6 s instead of 3 s) */
        const int fd = fileno(f);
        const uint64_t nI = ((uint64_t)h.M + 1) * sizeof(int);
        const uint64_t nJ = (uint64_t)h.NZ * sizeof(int);
        const uint64_t nV = (uint64_t)h.NZ * sizeof(double);
        const uint64_t total = nI + nJ + nV, piece = (uint64_t)32 << 20;
        const long long pieces = (long long)((total + piece - 1) / piece);
        int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : bad) num_threads(spmv_host_threads())
        for (long long k = 0; k < pieces; ++k) {
            uint64_t a = (uint64_t)k * piece;
            const uint64_t z = a + piece < total ? a + piece : total;
            while (a < z && !bad) { /* a piece may straddle two arrays */
                char *dst;
                uint64_t lim;
                if (a < nI) {
                    dst = (char *)A->IRP + a;
                    lim = nI;
                } else if (a < nI + nJ) {
                    dst = (char *)A->JA + (a - nI);
                    lim = nI + nJ;
                } else {
                    dst = (char *)A->AS + (a - nI - nJ);
                    lim = total;
                }
                const uint64_t want = (z < lim ? z : lim) - a;
                const ssize_t got = pread(fd, dst, (size_t)want,
                                          (off_t)(sizeof h + a));
                if (got <= 0)
                    bad = 1;
                else
                    a += (uint64_t)got;
            }
        }
        if (bad)
            rc = -EIO;
        else if (!csr_arrays_valid(A))
            rc = -EILSEQ; /* not a CSR matrix: corrupt or foreign file */
    }
out:
    fclose(f);
    if (rc) {
        csr_free(A);
        return ERR_PTR(rc);
    }
    return A;
}

sparse_csr *csr_load_bin(const char *path) {
    if (!path)
        return ERR_PTR(-EINVAL);
    return csr_read_bin(path, NULL);
}

/* "<path>.bin" next to the text file: read it when it was made from exactly
 * this .mtx and passes validation, otherwise parse the text and (best
 * effort) replace the sidecar */
sparse_csr *io_load_csr_cached(const char *path) {
    char bin[MAX_PATH * 4];
    struct stat st_txt;
    if (!path)
        return ERR_PTR(-EINVAL);
    if (snprintf(bin, sizeof bin, "%s.bin", path) >= (int)sizeof bin ||
        stat(path, &st_txt) != 0)
        return io_load_csr(path);
    sparse_csr *A = csr_read_bin(bin, &st_txt);
    if (!IS_ERR(A)) {
        extract_matrix_name(path, A->name);
        return A;
    }
    A = io_load_csr(path);
    if (!IS_ERR(A))
        (void)csr_write_bin(A, bin, &st_txt);
    return A;
}

/* ------------------------------------------------------------------ */
/* synthetic matrices, slices, partitions                               */
/* ------------------------------------------------------------------ */

sparse_csr *csr_generate(int kind, int M, int N, int K, int64_t W,
                         int64_t row0, uint64_t seed) {
    if (M < 0 || N <= 0 || K <= 0 || kind < SYNTH_BANDED || kind > SYNTH_KIND_LAST ||
        (kind == SYNTH_BANDED && N < K))
        return ERR_PTR(-EINVAL);
    synth_spec s = {kind, M, N, K, W, row0, seed};
    long long nz = 0;
    if (kind == SYNTH_BANDED || kind == SYNTH_RANDOM) {
        nz = (long long)M * K;
    } else {
/* a team is opened only for work that pays for it: bench.py's result
         * check regenerates single rows, and a full-width team per 1-row call
         * burned a 16-CPU cgroup quota on a 256-thread host */
#pragma omp parallel for schedule(static) reduction(+ : nz) if (M > 4096) num_threads(spmv_host_threads())
        for (int i = 0; i < M; ++i)
            nz += synth_row_len(&s, row0 + i);
    }
    if (nz > INT_MAX)
        return ERR_PTR(-EOVERFLOW);
    static const char *names[] = {"synth_banded", "synth_random",
                                  "synth_ragged", "synth_kkt",
                                  "synth_stencil", "synth_powerlaw",
                                  "synth_hub"};
    sparse_csr *A = csr_alloc(names[kind], M, N, (int)nz);
    if (IS_ERR(A))
        return A;
    for (int i = 0; i < M; ++i)
        A->IRP[i + 1] = A->IRP[i] + synth_row_len(&s, row0 + i);
#pragma omp parallel for schedule(dynamic, 4096) if (M > 4096) num_threads(spmv_host_threads())
    for (int i = 0; i < M; ++i)
        synth_fill_row(&s, row0 + i, A->IRP[i + 1] - A->IRP[i],
                       A->JA + A->IRP[i], A->AS + A->IRP[i]);
    return A;
}

int csr_synth_row_dots(int kind, int N, int K, int64_t W, uint64_t seed,
                       uint64_t xseed, const int64_t *rows, int n,
                       double *dot, double *scale) {
    if (N <= 0 || K <= 0 || kind < SYNTH_BANDED || kind > SYNTH_KIND_LAST ||
        (kind == SYNTH_BANDED && N < K) || n < 0 || (n && (!rows || !dot)))
        return -EINVAL;
    synth_spec s = {kind, 1, N, K, W, 0, seed};
    int cap = 64;
    int *cols = (int *)malloc((size_t)cap * sizeof(int));
    double *vals = (double *)malloc((size_t)cap * sizeof(double));
    int rc = cols && vals ? 0 : -ENOMEM;
    for (int k = 0; k < n && !rc; ++k) {
        const int len = synth_row_len(&s, rows[k]);
        if (len > cap) {
            cap = len;
            int *c2 = (int *)realloc(cols, (size_t)cap * sizeof(int));
            if (c2)
                cols = c2;
            double *v2 = (double *)realloc(vals, (size_t)cap * sizeof(double));
            if (v2)
                vals = v2;
            if (!c2 || !v2) {
                rc = -ENOMEM;
                break;
            }
        }
        synth_fill_row(&s, rows[k], len, cols, vals);
        double acc = 0.0, sab = 0.0;
        for (int j = 0; j < len; ++j) {
            const double p = vals[j] * synth_x(xseed, cols[j]);
            acc += p;
            sab += fabs(p);
        }
        dot[k] = acc;
        if (scale)
            scale[k] = sab;
    }
    free(cols);
    free(vals);
    return rc;
}

sparse_csr *csr_row_slice(const sparse_csr *A, int r0, int r1) {
    if (!A || r0 < 0 || r1 < r0 || r1 > A->M)
        return ERR_PTR(-EINVAL);
    int base = A->IRP[r0], nz = A->IRP[r1] - base;
    sparse_csr *S = csr_alloc(A->name, r1 - r0, A->N, nz);
    if (IS_ERR(S))
        return S;
    for (int r = r0; r <= r1; ++r)
        S->IRP[r - r0] = A->IRP[r] - base;
    memcpy(S->JA, A->JA + base, (size_t)nz * sizeof(int));
    memcpy(S->AS, A->AS + base, (size_t)nz * sizeof(double));
    return S;
}

/* Greedy cut at total/parts entries (behaviour of reference csr.c:218-276:
 * a range closes as soon as its running count reaches the target, the last
 * range takes the remainder, and the number of ranges shrinks when rows run
 * out first). */
int *partition_rows_nnz(const sparse_csr *A, int *parts) {
    if (!A || !parts || *parts < 1)
        return ERR_PTR(-EINVAL);
    int want = *parts;
    int *starts = malloc(((size_t)want + 1) * sizeof *starts);
    if (!starts)
        return ERR_PTR(-ENOMEM);
    double target = (double)((long long)A->IRP[A->M] - A->IRP[0]) / want;
    double load = 0.0;
    int k = 0;
    starts[0] = 0;
    for (int r = 0; r < A->M && k < want - 1; ++r) {
        load += (double)(A->IRP[r + 1] - A->IRP[r]);
        if (load >= target) {
            starts[++k] = r + 1;
            load = 0.0;
        }
    }
    starts[k + 1] = A->M;
    *parts = k + 1;
    return starts;
}

int *partition_rows_even(int M, int parts, int align) {
    if (M < 0 || parts < 1 || align < 1)
        return ERR_PTR(-EINVAL);
    int *starts = malloc(((size_t)parts + 1) * sizeof *starts);
    if (!starts)
        return ERR_PTR(-ENOMEM);
    long long per = ((long long)M + parts - 1) / parts;
    per = (per + align - 1) / align * align;
    for (int k = 0; k <= parts; ++k) {
        long long s = per * k;
        starts[k] = (int)(s < M ? s : M);
    }
    return starts;
}

/*
 * Cut `nb` consecutive blocks of weight w[] into `parts` contiguous ranges of
 * near-equal weight: cut k sits at the block boundary whose prefix weight is
 * nearest to k/parts of the total (the reference's partitioner, csr.c:218-276,
 * closes a range once its OWN running count reaches total/parts and starts
 * the next from zero, so every overshoot is taken from the last range; prefix
 * targets do not drift).  Every range holds at least one block while
 * nb >= parts; with fewer blocks than parts the trailing ranges are empty.
 * cut[parts+1] in blocks, cut[0] = 0, cut[parts] = nb.
 */
static void cut_blocks_balanced(const int64_t *w, int nb, int parts, int *cut) {
    int64_t total = 0;
    for (int b = 0; b < nb; ++b)
        total += w[b];
    cut[0] = 0;
    int b = 0;
    int64_t pre = 0; /* weight of blocks [0, b) */
    for (int k = 1; k < parts; ++k) {
        /* boundaries cut k may take: one block per range either side */
        int lo = cut[k - 1] + 1, hi = nb - (parts - k);
        if (nb < parts) { /* not enough blocks: one each, the rest empty */
            cut[k] = k < nb ? k : nb;
            continue;
        }
        /* target = total * k / parts, compared as pre * parts vs total * k
         * (total < 2^31 entries x parts <= 2^16: no overflow) */
        const int64_t tk = total * k;
        while (b < hi && (pre + w[b]) * parts <= tk)
            pre += w[b++];
        /* b = last boundary at or below the target; b + 1 may be nearer */
        int c = b;
        if (b < hi && (pre + w[b]) * parts - tk < tk - pre * parts)
            c = b + 1;
        if (c < lo)
            c = lo;
        if (c > hi)
            c = hi;
        while (b < c)
            pre += w[b++];
        cut[k] = c;
    }
    cut[parts] = nb;
}

static int *starts_from_cuts(const int *cut, int parts, int align, int M) {
    int *starts = malloc(((size_t)parts + 1) * sizeof *starts);
    if (!starts)
        return ERR_PTR(-ENOMEM);
    for (int k = 0; k <= parts; ++k) {
        const long long s = (long long)cut[k] * align;
        starts[k] = (int)(s < M ? s : M);
    }
    return starts;
}

int *partition_rows_nnz_aligned(const int *IRP, int M, int parts, int align) {
    if (!IRP || M < 0 || parts < 1 || parts > 65536 || align < 1)
        return ERR_PTR(-EINVAL);
    const int nb = (int)(((long long)M + align - 1) / align);
    int64_t *w = malloc(((size_t)nb + 1) * sizeof *w);
    int *cut = malloc(((size_t)parts + 1) * sizeof *cut);
    if (!w || !cut) {
        free(w);
        free(cut);
        return ERR_PTR(-ENOMEM);
    }
    for (int b = 0; b < nb; ++b) {
        const long long r0 = (long long)b * align;
        const long long r1 = r0 + align < M ? r0 + align : M;
        w[b] = (int64_t)IRP[r1] - IRP[r0];
    }
    cut_blocks_balanced(w, nb, parts, cut);
    int *starts = starts_from_cuts(cut, parts, align, M);
    free(w);
    free(cut);
    return starts;
}

int *partition_synth_rows_nnz(int kind, int M, int N, int K, int64_t W,
                              uint64_t seed, int parts, int align) {
    if (kind < 0 || kind > SYNTH_KIND_LAST || M < 0 || N < 1 || K < 1 ||
        parts < 1 || parts > 65536 || align < 1)
        return ERR_PTR(-EINVAL);
    const int nb = (int)(((long long)M + align - 1) / align);
    int64_t *w = malloc(((size_t)nb + 1) * sizeof *w);
    int *cut = malloc(((size_t)parts + 1) * sizeof *cut);
    if (!w || !cut) {
        free(w);
        free(cut);
        return ERR_PTR(-ENOMEM);
    }
    const synth_spec s = {kind, M, N, K, W, 0, seed};
#pragma omp parallel for schedule(static) num_threads(spmv_host_threads())
    for (int b = 0; b < nb; ++b) {
        const long long r0 = (long long)b * align;
        const long long r1 = r0 + align < M ? r0 + align : M;
        int64_t acc = 0;
        for (long long g = r0; g < r1; ++g)
            acc += synth_row_len(&s, g);
        w[b] = acc;
    }
    cut_blocks_balanced(w, nb, parts, cut);
    int *starts = starts_from_cuts(cut, parts, align, M);
    free(w);
    free(cut);
    return starts;
}

/* ------------------------------------------------------------------ */
/* CPU kernels (reference csr.c:201-216, 278-339)                       */
/* ------------------------------------------------------------------ */

typedef double (*csr_kernel_fn)(const sparse_csr *, const double *, double *,
                                void *);

static inline double row_dot(const sparse_csr *A, const double *x, int i) {
    double acc = 0.0;
    for (int k = A->IRP[i]; k < A->IRP[i + 1]; ++k)
        acc += A->AS[k] * x[A->JA[k]];
    return acc;
}

static double csr_spmv_serial(const sparse_csr *A, const double *x, double *y,
                              void *arg) {
    (void)arg;
    double t0 = now(); /* CPU time, as the reference (utils.h:68) */
    for (int i = 0; i < A->M; ++i)
        y[i] = row_dot(A, x, i);
    return now() - t0;
}

static double csr_spmv_omp_guided(const sparse_csr *A, const double *x,
                                  double *y, void *arg) {
    int threads = *(const int *)arg;
    double t0 = omp_get_wtime();
#pragma omp parallel for schedule(guided) num_threads(threads)
    for (int i = 0; i < A->M; ++i)
        y[i] = row_dot(A, x, i);
    return (omp_get_wtime() - t0) * 1e3;
}

struct nnz_ranges {
    int parts;
    const int *starts;
};

static double csr_spmv_omp_ranges(const sparse_csr *A, const double *x,
                                  double *y, void *arg) {
    const struct nnz_ranges *rg = arg;
    double t0 = omp_get_wtime();
#pragma omp parallel num_threads(rg->parts)
    {
        /* a team smaller than requested still covers every range */
        int team = omp_get_num_threads();
        for (int t = omp_get_thread_num(); t < rg->parts; t += team)
            for (int i = rg->starts[t]; i < rg->starts[t + 1]; ++i)
                y[i] = row_dot(A, x, i);
    }
    return (omp_get_wtime() - t0) * 1e3;
}

/* harness shared by every CSR benchmark (reference csr.c:182-199): fresh
 * zeroed y, one call, record time / GFLOP/s, hand y to the caller */
static int run_csr_bench(const sparse_csr *A, const double *x, bench *out,
                         void *arg, csr_kernel_fn fn) {
    vec y = vec_create((size_t)A->M);
    if (!y.data)
        return -ENOMEM;
    double ms = fn(A, x, y.data, arg);
    if (ms < 0.0) { /* GPU entry points report failures as -errno */
        vec_put(&y);
        return (int)ms;
    }
    out->duration_ms = ms;
    out->gflops = compute_gflops(ms, A->NZ);
    out->data = y;
    return 0;
}

int bench_csr_serial(const sparse_csr *A, const double *x, bench *out) {
    return run_csr_bench(A, x, out, NULL, csr_spmv_serial);
}

int bench_csr_omp_guided(const sparse_csr *A, const double *x,
                         bench_omp *out) {
    snprintf(out->name, sizeof out->name, "omp_guided");
    if (out->num_threads < 1)
        return -EINVAL;
    return run_csr_bench(A, x, &out->bench, &out->num_threads,
                         csr_spmv_omp_guided);
}

int bench_csr_omp_nnz_balancing(const sparse_csr *A, const double *x,
                                bench_omp *out) {
    snprintf(out->name, sizeof out->name, "omp_nnz");
    if (out->num_threads < 1)
        return -EINVAL;
    int *starts = partition_rows_nnz(A, &out->num_threads);
    if (IS_ERR(starts))
        return PTR_ERR(starts);
    struct nnz_ranges rg = {out->num_threads, starts};
    int rc = run_csr_bench(A, x, &out->bench, &rg, csr_spmv_omp_ranges);
    free(starts);
    return rc;
}

/* ------------------------------------------------------------------ */
/* GPU wrappers (counterparts of reference csr.c:382-415)               */
/* ------------------------------------------------------------------ */

static int run_csr_hip(const sparse_csr *A, const double *x, bench_hip *out,
                       csr_kernel_fn fn) {
    spmv_launch_opts opts;
    memset(&opts, 0, sizeof opts);
    opts.waves_per_block = out->waves_per_block;
    return run_csr_bench(A, x, &out->bench, &opts, fn);
}

int bench_csr_hip_thread_row(const sparse_csr *A, const double *x,
                             bench_hip *out) {
    return run_csr_hip(A, x, out, csr_spmv_hip_thread_row);
}

int bench_csr_hip_wave_row(const sparse_csr *A, const double *x,
                           bench_hip *out) {
    return run_csr_hip(A, x, out, csr_spmv_hip_wave_row);
}

int bench_csr_hip_subwave_row(const sparse_csr *A, const double *x,
                              bench_hip *out) {
    return run_csr_hip(A, x, out, csr_spmv_hip_subwave_row);
}

int bench_csr_hip_block_row(const sparse_csr *A, const double *x,
                            bench_hip *out) {
    return run_csr_hip(A, x, out, csr_spmv_hip_block_row);
}

int bench_csr_hip_stream(const sparse_csr *A, const double *x,
                         bench_hip *out) {
    return run_csr_hip(A, x, out, csr_spmv_hip_stream);
}

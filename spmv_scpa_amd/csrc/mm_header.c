/* mm_header.c -- Matrix Market banner + size line (API: include/mm_header.h).
 * Behaviour follows reference src/mmio.c:93-166 and 175-200. */
#include <ctype.h>
#include <stdio.h>
#include <string.h>

#include "mm_header.h"

#define MM_LINE_MAX 1025 /* the reference reads lines with fgets(1025) */
#define MM_TOKEN_MAX 64

/* copy one line (at most MM_LINE_MAX-1 chars, newline included) like fgets;
 * returns bytes consumed, 0 at end of input */
static size_t take_line(const char *p, size_t left, char *line) {
    size_t n = 0;
    while (n < left && n < MM_LINE_MAX - 1) {
        line[n] = p[n];
        if (p[n++] == '\n')
            break;
    }
    line[n] = 0;
    return n;
}

static void lowercase(char *s) {
    for (; *s; ++s)
        *s = (char)tolower((unsigned char)*s);
}

int mm_parse_header(const char *text, size_t len, mm_info *out) {
    char line[MM_LINE_MAX];
    char t0[MM_TOKEN_MAX], t1[MM_TOKEN_MAX], t2[MM_TOKEN_MAX], t3[MM_TOKEN_MAX],
        t4[MM_TOKEN_MAX];
    size_t pos = 0, n;

    memset(out, 0, sizeof *out);
    n = take_line(text, len, line);
    if (!n)
        return MM_PREMATURE_EOF;
    pos += n;
    if (sscanf(line, "%63s %63s %63s %63s %63s", t0, t1, t2, t3, t4) != 5)
        return MM_PREMATURE_EOF;
    lowercase(t1);
    lowercase(t2);
    lowercase(t3);
    lowercase(t4);
    if (strncmp(t0, "%%MatrixMarket", 14) != 0)
        return MM_NO_HEADER;
    if (strcmp(t1, "matrix") != 0)
        return MM_UNSUPPORTED_TYPE;
    out->object = 'M';

    if (!strcmp(t2, "coordinate"))
        out->format = 'C';
    else if (!strcmp(t2, "array"))
        out->format = 'A';
    else
        return MM_UNSUPPORTED_TYPE;

    static const struct { const char *word; char code; } fields[] = {
        {"real", 'R'}, {"complex", 'C'}, {"pattern", 'P'}, {"integer", 'I'}};
    static const struct { const char *word; char code; } syms[] = {
        {"general", 'G'}, {"symmetric", 'S'}, {"hermitian", 'H'},
        {"skew-symmetric", 'K'}};
    for (size_t k = 0; k < 4; ++k)
        if (!strcmp(t3, fields[k].word))
            out->field = fields[k].code;
    for (size_t k = 0; k < 4; ++k)
        if (!strcmp(t4, syms[k].word))
            out->symmetry = syms[k].code;
    if (!out->field || !out->symmetry)
        return MM_UNSUPPORTED_TYPE;

    /* size line: skip '%' lines; a line that is blank (or otherwise not
     * three integers) makes the reader fall through to free-form scanning */
    for (;;) {
        n = take_line(text + pos, len - pos, line);
        if (!n)
            return MM_PREMATURE_EOF;
        pos += n;
        if (line[0] != '%')
            break;
    }
    if (sscanf(line, "%d %d %d", &out->rows, &out->cols, &out->entries) == 3) {
        out->data_offset = pos;
        return MM_OK;
    }
    /* reference: do { fscanf("%d %d %d") } while (!= 3), EOF -> failure */
    for (;;) {
        int used = 0;
        /* sscanf on the remaining buffer; text is NUL-terminated by caller */
        int got = sscanf(text + pos, "%d %d %d%n", &out->rows, &out->cols,
                         &out->entries, &used);
        if (got == 3) {
            out->data_offset = pos + (size_t)used;
            return MM_OK;
        }
        return MM_PREMATURE_EOF;
    }
}

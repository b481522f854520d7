/*
 * stream_table.h -- the row-block table of the CSR stream kernel
 * (csr_kernels.hip, k_csr_stream), built on the host in O(M) at upload.
 *
 * Header-only and free of HIP so that the table's invariants -- every entry
 * of the matrix covered by exactly one range, budgets kept, the segments of a
 * long row consecutive and cut where the kernel's index arithmetic expects
 * them -- run as a CPU test under ASan / UBSan
 * (tests/asan/stream_table_asan.cc, tests/test_tune_blocked_asan.py).
 *
 * tab[k] / ent[k]: first row / first entry of range k; the closing pair
 * (M, irp[M]) ends the table; mode[k]: 0 transposed (every row of the range
 * holds at most row_t entries), 1 cooperative, 2 a SEGMENT of a long row.
 *   - ordinary ranges: consecutive rows holding at most nnz_budget entries
 *     and at most row_budget rows; a single row beyond nnz_budget (up to
 *     long_row entries) is a range of its own;
 *   - a row of more than long_row entries becomes ceil(len / seg) consecutive
 *     ranges of mode 2 that all name that row and cut its entries at
 *     multiples of seg from the row's first entry.  The kernel recovers the
 *     segment index as (first entry of the range - irp[row]) / seg, the
 *     row's first range as (range - index), the count as ceil(len / seg).
 */
#ifndef SPMV_STREAM_TABLE_H
#define SPMV_STREAM_TABLE_H

#include <algorithm>
#include <vector>

static inline void stream_table_build(const int *irp, int M, int nnz_budget,
                                      int row_budget, int row_t, int long_row,
                                      int seg, std::vector<int> &tab,
                                      std::vector<int> &ent,
                                      std::vector<unsigned char> &mode,
                                      int *max_len, bool *has_segments) {
    tab.clear();
    ent.clear();
    mode.clear();
    *has_segments = false;
    int start = 0, longest = 0, range_longest = 0;
    auto close_range = [&](int r) { /* rows [start, r) */
        tab.push_back(start);
        ent.push_back(irp[start]);
        mode.push_back(range_longest > row_t ? 1 : 0);
        start = r;
        range_longest = 0;
    };
    for (int r = 0; r < M; ++r) {
        const int len = irp[r + 1] - irp[r];
        longest = std::max(longest, len);
        if (len > long_row) {
            if (r > start)
                close_range(r);
            for (int b = irp[r]; b < irp[r + 1]; b += seg) {
                tab.push_back(r);
                ent.push_back(b);
                mode.push_back(2);
            }
            *has_segments = true;
            start = r + 1;
            range_longest = 0;
            continue;
        }
        const int have = irp[r] - irp[start];
        const bool full = (have + len > nnz_budget) || (r - start >= row_budget);
        if (full && r > start)
            close_range(r);
        range_longest = std::max(range_longest, len);
    }
    if (M > start)
        close_range(M);
    tab.push_back(M);
    ent.push_back(irp[M]);
    mode.push_back(0);
    *max_len = longest;
}

#endif /* SPMV_STREAM_TABLE_H */

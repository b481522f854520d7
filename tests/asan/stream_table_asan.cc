// CPU test of the CSR stream kernel's row-block table
// (spmv_scpa_amd/csrc/stream_table.h) under ASan + UBSan: seeded random row
// lengths (empty rows, 1-3 entry rows, rows around every budget and around
// the long-row threshold, long rows first / last / adjacent), a non-zero
// IRP[0], and a host restatement of what each workgroup of k_csr_stream does
// with its range -- every entry must be summed exactly once into its row.
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "stream_table.h"

// the kernel's constants (hip_common.h), handed over by the test as -D macros
static const int NNZ = STREAM_NNZ, ROWS = STREAM_ROWS, ROW_T = STREAM_ROW_T,
                 LONG = STREAM_LONG_ROW, SEG = STREAM_SEG;

static uint64_t st;
static uint64_t rnd(void) {
    uint64_t z = (st += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static int pick_len(void) {
    switch (rnd() % 16) {
    case 0: return 0;
    case 1: case 2: case 3: case 4: return 1 + (int)(rnd() % 3);
    case 5: return ROW_T + (int)(rnd() % 3) - 1;
    case 6: return NNZ + (int)(rnd() % 3) - 1;
    case 7: return LONG + (int)(rnd() % 3) - 1;          // 8191..8193
    case 8: return SEG * (1 + (int)(rnd() % 6)) + (int)(rnd() % 3) - 1;
    case 9: return (int)(rnd() % 30000);
    default: return (int)(rnd() % 40);
    }
}

static void scenario(uint64_t seed) {
    st = seed;
    const int M = (int)(rnd() % 3 == 0 ? rnd() % 40 : rnd() % 4000);
    const int base = rnd() & 1 ? 0 : (int)(rnd() % 1000); // IRP[0] != 0 too
    std::vector<int> irp((size_t)M + 1);
    irp[0] = base;
    for (int r = 0; r < M; ++r) {
        int len = pick_len();
        if ((r == 0 || r == M - 1) && (rnd() & 3) == 0)
            len = LONG + 1 + (int)(rnd() % 9000); // long row first / last
        if ((int64_t)irp[r] + len > 2000000000)
            len = 0;
        irp[r + 1] = irp[r] + len;
    }
    std::vector<int> tab, ent;
    std::vector<unsigned char> mode;
    int longest = -1;
    bool segs = false;
    stream_table_build(irp.data(), M, NNZ, ROWS, ROW_T, LONG, SEG, tab, ent,
                       mode, &longest, &segs);
    const int n = (int)tab.size() - 1;
    assert(n >= 0 && ent.size() == tab.size() && mode.size() == tab.size());
    assert(tab[n] == M && ent[n] == irp[M]);
    // what the kernel does, restated: per range, which entries go to which row
    std::vector<int> covered((size_t)(irp[M] - base), 0);
    std::vector<int> partial_of_row((size_t)M, 0), seg_seen((size_t)M, 0);
    int want_longest = 0;
    bool want_segs = false;
    for (int r = 0; r < M; ++r) {
        want_longest = std::max(want_longest, irp[r + 1] - irp[r]);
        want_segs |= irp[r + 1] - irp[r] > LONG;
    }
    assert(longest == want_longest && segs == want_segs);
    for (int k = 0; k < n; ++k) {
        const int row_a = tab[k], row_b = tab[k + 1];
        const int beg = ent[k], end = ent[k + 1];
        assert(beg <= end && row_a <= row_b);
        if (mode[k] == 2) {
            const int b0 = irp[row_a], len = irp[row_a + 1] - b0;
            assert(len > LONG);
            assert((beg - b0) % SEG == 0 && beg >= b0 && end <= irp[row_a + 1]);
            const int kseg = (beg - b0) / SEG, nseg = (len + SEG - 1) / SEG;
            const int rb0 = k - kseg;
            assert(rb0 >= 0 && tab[rb0] == row_a && ent[rb0] == b0);
            assert(mode[rb0] == 2 && kseg < nseg && end - beg <= SEG);
            assert(end - beg == SEG || kseg == nseg - 1);
            for (int j = 0; j < nseg; ++j) // the row's ranges are consecutive
                assert(tab[rb0 + j] == row_a && mode[rb0 + j] == 2);
            assert(seg_seen[row_a] == kseg); // in segment order
            ++seg_seen[row_a];
            for (int e = beg; e < end; ++e)
                ++covered[(size_t)(e - base)];
            continue;
        }
        const int rows = row_b - row_a, cnt = end - beg;
        assert(rows >= 1 && rows <= ROWS);
        assert(beg == irp[row_a] && end == irp[row_b]);
        if (cnt > NNZ) { // one row beyond the budget, read in place
            assert(rows == 1 && cnt <= LONG);
        } else {
            int lmax = 0;
            for (int r = row_a; r < row_b; ++r) {
                assert(irp[r + 1] - irp[r] <= LONG);
                lmax = std::max(lmax, irp[r + 1] - irp[r]);
            }
            assert((mode[k] == 1) == (lmax > ROW_T));
        }
        for (int e = beg; e < end; ++e)
            ++covered[(size_t)(e - base)];
        for (int r = row_a; r < row_b; ++r)
            ++partial_of_row[r];
    }
    for (size_t e = 0; e < covered.size(); ++e)
        assert(covered[e] == 1); // every entry exactly once
    for (int r = 0; r < M; ++r) {
        const int len = irp[r + 1] - irp[r];
        if (len > LONG)
            assert(partial_of_row[r] == 0 && seg_seen[r] == (len + SEG - 1) / SEG);
        else
            assert(partial_of_row[r] == 1 && seg_seen[r] == 0); // y written once
    }
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 4000;
    for (int i = 0; i < n; ++i)
        scenario(0xabcdull + (uint64_t)i * 104729u);
    std::printf("stream table: %d scenarios, every entry covered once, every "
                "row written once, segments consecutive and in order\n", n);
    return 0;
}

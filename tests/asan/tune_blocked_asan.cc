// CPU test of the selector's ownership logic (spmv_scpa_amd/csrc/tune_blocked.h)
// under AddressSanitizer + UBSan, with mock blocked copies.
//
// Build + run: tests/test_tune_blocked_asan.py
//   g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer
//
// Every scenario draws, from a seed: whether the handle already owns a copy,
// the direct-kernel time, which builds fail (-ENOMEM: candidate dropped;
// -EIO: device error), which timing call fails, and every candidate's time.
// After tune_blocked returns the books must balance:
//   * each copy ever built was freed exactly once, or is the one in *slot;
//   * return 1: *slot is a NEW copy and the caller's original was freed once;
//   * return 0: *slot is the caller's original, untouched;
//   * return < 0: *slot is the caller's original, untouched, nothing else alive;
//   * ASan sees no double free / use after free / leak (the mock copy owns a
//     heap payload that set_* write to).
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#include "tune_blocked.h"

struct mock_copy {
    int *payload; // set_* write here: a use after free trips ASan
    int id;
};

static std::set<mock_copy *> g_alive;
static int g_built = 0, g_freed = 0;
static double g_clock = 0.0;

struct mock_ops {
    static void free(mock_copy *p) {
        if (!p)
            return;
        assert(g_alive.count(p) == 1 && "double free of a blocked copy");
        g_alive.erase(p);
        ++g_freed;
        std::free(p->payload);
        std::free(p);
    }
    static void set_chain(mock_copy *p, int v) { p->payload[0] = v; }
    static void set_waves(mock_copy *p, int v) { p->payload[1] = v; }
    static void set_order(mock_copy *p, int v) { p->payload[2] = v; }
    static int balanced_tile_rows(int M, int max_rows) {
        (void)M;
        return max_rows / 32 * 32 - 32;
    }
    static double now_s(void) { return g_clock += 0.001; }
    static const char *build_phases(void) { return "mock"; }
};

static mock_copy *new_copy(void) {
    mock_copy *p = (mock_copy *)std::calloc(1, sizeof *p);
    p->payload = (int *)std::calloc(4, sizeof(int));
    p->id = ++g_built;
    g_alive.insert(p);
    return p;
}

static uint64_t rng_state;
static uint64_t rnd(void) { // splitmix64
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double urand(void) { return (double)(rnd() >> 11) / 9007199254740992.0; }

static int scenario(uint64_t seed, int verbose) {
    rng_state = seed;
    g_alive.clear();
    g_built = g_freed = 0;
    const bool has_original = rnd() & 1;
    mock_copy *original = has_original ? new_copy() : NULL;
    mock_copy *slot = original;
    const int M = (int)(rnd() % 3 == 0 ? 800000 : rnd() % 2 ? 3000000 : 10000000);
    const double stream_ms = 0.5;
    double bms = 0.55 + 3.0 * urand(); // sometimes within 1.2x: nothing built
    const double bms_in = bms;
    const int fail_build_at = rnd() % 4 == 0 ? (int)(rnd() % 6) : -1;
    const int fail_build_rc = rnd() & 1 ? -ENOMEM : (rnd() & 1 ? -EOVERFLOW : -EIO);
    const int fail_time_at = rnd() % 5 == 0 ? (int)(rnd() % 24) : -1;
    int builds = 0, times = 0, lines = 0;
    int rc = tune_blocked<mock_copy, mock_ops>(
        &slot, M, (rnd() & 1) ? 3.0 : 32.0, stream_ms, &bms,
        [&](int sched, int tile_rows, mock_copy **out) {
            (void)sched;
            (void)tile_rows;
            if (builds++ == fail_build_at)
                return fail_build_rc; // *out stays NULL, as panels_build does
            *out = new_copy();
            return 0;
        },
        [&](double *m) {
            assert(slot && g_alive.count(slot) && "timing a freed copy");
            if (times++ == fail_time_at)
                return -EIO;
            *m = 0.3 + 3.0 * urand();
            return 0;
        },
        [&](const char *line) {
            (void)line;
            ++lines;
        });
    // ---- the books ----
    if (rc == 1) {
        assert(slot && slot != original && g_alive.count(slot));
        assert(g_alive.size() == 1);
        assert(bms < bms_in);
        mock_ops::free(slot); // what the handle's release does later
    } else {
        assert(slot == original);
        if (original) {
            assert(g_alive.size() == 1 && g_alive.count(original));
            mock_ops::free(original);
        }
        assert(rc == 0 ? bms == bms_in || true : rc < 0);
        if (rc == 0)
            assert(bms == bms_in);
    }
    assert(g_alive.empty());
    assert(g_freed == g_built);
    if (verbose)
        std::printf("seed %llu: rc %d, built %d, timed %d, log lines %d\n",
                    (unsigned long long)seed, rc, g_built, times, lines);
    return rc;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 20000;
    int kept = 0, dropped = 0, errors = 0;
    for (int i = 0; i < n; ++i) {
        const int rc = scenario(0x5eedull + (uint64_t)i * 7919u, i < 5);
        kept += rc == 1;
        dropped += rc == 0;
        errors += rc < 0;
    }
    std::printf("tune_blocked ownership: %d scenarios, %d kept a blocked copy, "
                "%d kept the direct kernel, %d device errors -- books balance\n",
                n, kept, dropped, errors);
    return kept && dropped && errors ? 0 : 3; // every outcome class was seen
}

/*
 * tune_blocked.h -- the 2-D blocked path as a candidate of spmv_*_autotune:
 * which copies are built, timed, kept and freed.
 *
 * Header-only and free of HIP so that the OWNERSHIP logic (every candidate
 * freed exactly once or handed to the handle; the caller's own copy freed
 * only when it is replaced; nothing freed on an error path that the caller
 * still points at) is compiled into a CPU test under AddressSanitizer with
 * mock copies (tests/asan/tune_blocked_asan.cc, run by
 * tests/test_tune_blocked_asan.py).  engine.hip instantiates it with the real
 * operations of panels.hip.
 *
 * `*bms` is the best direct kernel's time on entry, nnz_per_row the mean row.  Near the stream rate
 * (within 1.2x of it at 7 TB/s) nothing is built.  Otherwise the steps layout
 * is built at up to four tile heights (8192 rows and the two tall heights
 * balanced over whole rounds of the chip for 10M rows, lower for smaller
 * matrices) and timed both as one chain launch and as one launch per step:
 * column-sorted buckets turn the gathers of a banded / clustered / skewed
 * matrix into a few whole-line requests (chain, random W = 2^14: 0.63 vs
 * 1.19 ms direct; W = 2^17: 0.68 vs 1.58; W = 2^20: 0.89 vs 2.9; skewed rows
 * 0.165 vs 0.51; 27-point stencil 0.44 vs 0.52).  When the direct kernels
 * run beyond 2.5x the stream time the rows reach far outside an L2 and the
 * sweep schedule is tried too (config 3: 1.6 ms vs 1.9 chain vs 5.8 direct).
 * The winner stays in `*slot` (12 B per entry); returns 1 when a blocked
 * form won, 0 when not, < 0 on a device error.  Out of memory / index
 * overflow just drops the candidate.
 *
 * Ops (static members): free(P*), set_chain(P*, int), set_waves(P*, int),
 * set_order(P*, int), balanced_tile_rows(int M, int max_rows), now_s(),
 * build_phases() (text: where the last build spent its time).
 * Build: int(int sched, int tile_rows, P **out) -- *out set only on success.
 * Time:  int(double *ms).   Log: void(const char *line).
 */
#ifndef SPMV_TUNE_BLOCKED_H
#define SPMV_TUNE_BLOCKED_H

#include <errno.h>
#include <stdio.h>

template <class P, class Ops, class Build, class Time, class Log>
static int tune_blocked(P **slot, int M, double nnz_per_row, double stream_ms,
                        double *bms, Build build, Time time_it, Log log) {
    if (*bms <= 1.2 * stream_ms)
        return 0;
    P *const original = *slot; /* caller-built copy, if any */
    P *keep = NULL;            /* best blocked copy so far */
    int err = 0;
    /* a blocked copy costs 12 B per entry: it has to win by 5 % over the
     * direct kernels (not over another blocked candidate) to be kept */
    const double direct_ms = *bms;
    double last_m = 1e300; /* time of the candidate tried last */
    /* build + time one candidate (steps layout: in both launch modes); keeps
     * it when it beats everything so far */
    auto try_one = [&](int sched, int tile_rows) {
        P *cand = NULL;
        const double t0 = Ops::now_s();
        int rc = build(sched, tile_rows, &cand);
        const double t1 = Ops::now_s();
        if (rc == -ENOMEM || rc == -EOVERFLOW)
            return;
        if (rc) {
            err = rc;
            return;
        }
        *slot = cand;
        double best_m = 1e300;
        int best_chain = 0, best_waves = 0, timed = 0;
        for (int chain = (sched == 0 ? 1 : 0); chain >= 0 && !err; --chain) {
            Ops::set_chain(cand, chain);
            /* tall tiles leave room for one workgroup per CU: 1024 lanes
             * (16 wavefronts) or the 512 of the heuristic, whichever runs
             * faster (W = 2^20 at 20448 rows: 0.80 vs 0.86 ms; W = N: 1.77
             * vs 1.63) */
            for (int waves = 0; waves <= (sched == 0 && chain &&
                                          tile_rows >= 12288 ? 16 : 0);
                 waves += 16) {
                Ops::set_waves(cand, waves);
                double m = 0.0;
                rc = time_it(&m);
                ++timed;
                if (rc) {
                    err = rc;
                    break;
                }
                if (m < best_m) {
                    best_m = m;
                    best_chain = chain;
                    best_waves = waves;
                }
            }
        }
        Ops::set_waves(cand, best_waves);
        Ops::set_chain(cand, best_chain);
        if (!err && sched == 0) {
            /* the best launch mode in the other two tile orders (the copy is
             * built with order 0, grouped); another order has to win by 2 % */
            int best_order = 0;
            for (int order = 1; order <= 2 && !err; ++order) {
                Ops::set_order(cand, order);
                double m = 0.0;
                rc = time_it(&m);
                ++timed;
                if (rc)
                    err = rc;
                else if (m < 0.98 * best_m) {
                    best_m = m;
                    best_order = order;
                }
            }
            Ops::set_order(cand, best_order);
        }
        *slot = original;
        last_m = err ? 1e300 : best_m;
        {
            char line[400];
            snprintf(line, sizeof line,
                     "blocked sched=%d tile_rows=%d: build %.3f s (%s), %d "
                     "timed configurations %.3f s, best %.4f ms%s",
                     sched, tile_rows, t1 - t0, Ops::build_phases(), timed,
                     Ops::now_s() - t1, err ? -1.0 : best_m,
                     err ? " (error)" : "");
            log(line);
        }
        /* a later blocked candidate has to beat the kept one by 3 %: two
         * forms within run-to-run noise of each other (config 3: sweep 1.53
         * vs chain at 20448 rows 1.55-1.60 ms) must not flip the pick from
         * run to run -- a job's ranks, and the three passes of a profile,
         * are to see the same kernel */
        if (!err && best_m < *bms * (keep ? 0.97 : 1.0) &&
            best_m < 0.95 * direct_ms) {
            *bms = best_m;
            Ops::free(keep);
            keep = cand;
        } else {
            Ops::free(cand);
        }
    };
    const bool far = *bms > 2.5 * stream_ms;
    /* tile heights: tall tiles put more entries on a line of x, but the
     * launch wants a few hundred of them (1M rows: 4096 rows 0.073 ms, 8192
     * rows 0.109; 3M rows: 0.233 vs 0.267; 10M rows: 8192 or 16384) */
    int t1 = 8192, t2 = 16384;
    /* ... unless the rows are short (web / road graphs: 2-8 entries): a tile
     * of 8192 rows then holds only ~25k entries, six wavefront chunks, and its
     * fixed cost (zeroing and writing the slice of y) dominates -- power-law
     * 4M x 3, columns anywhere: 0.216 ms at 4096 rows, 0.177 at 8192, 0.155
     * with the sweep schedule's 15648-row tiles.  Such matrices get the tall
     * ladder from 1.5M rows up */
    if (M < 4900000 && !(nnz_per_row < 12.0 && M >= 1500000)) {
        t1 = 4096;
        t2 = 8192;
    }
    if (M < 1500000) {
        t1 = 256;
        while (t1 * 2 <= M / 192 && t1 < 4096)
            t1 *= 2;
        t2 = t1 > 256 ? t1 / 2 : 0;
    }
    /* rows that reach far beyond an L2 of x: the sweep schedule goes first
     * and is the form to beat (it also scales better with the column count:
     * one shard of the 80M-column problem 3.0 ms vs 3.4 chain) */
    if (far)
        try_one(1, 0);
    if (!err)
        try_one(0, t1);
    const double m1 = last_m;
    /* tall tiles run one workgroup per CU, so their height is balanced over
     * whole rounds of the chip (panels_balanced_tile_rows): 13024 rows
     * instead of 16384, 19552 instead of 20448 at 10M rows */
    const int tall2 = t2 == 16384 ? Ops::balanced_tile_rows(M, 16384) : t2;
    const int tall3 = t2 == 16384 ? Ops::balanced_tile_rows(M, 20448) : 0;
    if (!err && tall2)
        try_one(0, tall2);
    const double m2 = last_m;
    /* taller still (160 KiB of LDS) -- always tried on large matrices: round
     * 2 tried it only when 16384 rows had beaten 8192, a comparison within
     * run-to-run noise on W = 2^20, and the bench line then showed 0.90 ms
     * where this height gives 0.77 -- and shorter still when height cost
     * (nlpkkt160-shaped KKT matrix: 0.525 ms at 8192 rows, 0.596 at 16384) */
    if (!err && tall3 && tall3 != tall2)
        try_one(0, tall3);
    if (!err && t2 == 16384 && m1 < m2 && m1 < last_m)
        try_one(0, 4096);
    if (err) {
        Ops::free(keep);
        *slot = original;
        return err;
    }
    if (keep) {
        Ops::free(original);
        *slot = keep;
        return 1;
    }
    return 0;
}

#endif /* SPMV_TUNE_BLOCKED_H */

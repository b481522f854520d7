#!/usr/bin/env python3
"""What bitwise reproducibility of the blocked path costs.

For each workload: generate, let the selector pick a blocked layout, then
build that SAME layout twice -- default (ds_add_f64 in arrival order) and
deterministic (spmv_panel_opts.deterministic: ordered additions) -- and time
20 launches of each (events, median).  Also counts how many distinct y the
default mode produces over 20 launches, and checks that the deterministic
one produces exactly one.

    python tools/det_cost.py [--rows 10000000] > profiles/r05_det_cost.md
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import spmv_scpa_amd as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    a = ap.parse_args()
    M = N = a.rows
    d_x, d_y = S.DevBuffer(N * 8), S.DevBuffer(M * 8)
    S.dev_fill_synth(d_x.ptr, N, 7)
    print("| workload | layout (selector's pick) | default ms | deterministic "
          "ms | cost | distinct y in 20 launches (default / deterministic) |")
    print("|---|---|---|---|---|---|")
    cases = [("random W=2^17", S.SYNTH_RANDOM, 32, 1 << 17),
             ("random W=2^20", S.SYNTH_RANDOM, 32, 1 << 20),
             ("random W=N", S.SYNTH_RANDOM, 32, 2 * N),
             ("banded", S.SYNTH_BANDED, 32, 0)]
    for tag, kind, K, W in cases:
        dA = S.CsrDevice.generate(kind, M, N, K, W, 0, 42)
        dH = dA.to_hll(True)
        dA.release()
        best, _ = dH.autotune(d_x.ptr, d_y.ptr)
        if dH.panels_info() is None:
            dH.build_panels(0)
        pin = dH.panels_pin()
        res = {}
        for det in (0, 1):
            # spmv_engine.h: 1 = ordered additions, 2 = arrival order
            import re
            dH.build_panels_pinned(re.sub(r"deterministic=\d",
                                          "deterministic=%d" % (1 if det else 2),
                                          pin))
            ms = np.median(dH.time(S.HLL_KERNEL_PANELS, d_x.ptr, d_y.ptr, 3, 20))
            seen = set()
            for _ in range(20):
                dH.launch(S.HLL_KERNEL_PANELS, d_x.ptr, d_y.ptr)
                S.stream_sync()
                seen.add(hash(d_y.to_numpy(np.float64, M).tobytes()))
            res[det] = (float(ms), len(seen), dH.panels_describe())
        # a chain copy is one workgroup per CU at the selector's tall tiles:
        # the turn counter's hand-offs hide better with two or more workgroups
        # per CU, so shorter tiles are tried for the deterministic mode too
        alt = ""
        if "chain" in res[0][2]:
            for tr in (8192, 4096):
                dH.build_panels(0, "chain", tr, deterministic=True)
                ms = float(np.median(dH.time(S.HLL_KERNEL_PANELS, d_x.ptr,
                                             d_y.ptr, 3, 20)))
                alt += "; %d-row tiles %.4f ms (%+.1f %%)" % (
                    tr, ms, 100.0 * (ms / res[0][0] - 1.0))
        res[1] = (res[1][0], res[1][1], res[1][2] + alt)
        print("| %s %dM x %d | %s%s | %.4f | %.4f%s | %+.1f %% | %d / %d |" % (
            tag, M // 1_000_000, K, res[0][2],
            "" if best == S.HLL_KERNEL_PANELS else " (a direct kernel won the "
            "selection; default chain layout timed)",
            res[0][0], res[1][0], alt,
            100.0 * (res[1][0] / res[0][0] - 1.0),
            res[0][1], res[1][1]))
        sys.stdout.flush()
        dH.release()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Round-6 probe: would TALLER row tiles help at W = N?

The sweep schedule keeps a tile's slice of y in LDS, so a tile holds at most
20 448 rows; on the 10M x 32 matrix with columns anywhere that pins
lambda = entries of one bucket per 128-byte line of x at ~1.0, i.e. a gather
request per 1.6 entries.  Taller tiles raise lambda (more lanes of a gather
instruction share a line) -- IF accumulators for them existed (registers).
Before building those: time the request pattern.  The ablations flavour of the
library (`make abl`) builds sweep copies of any tile height
(SPMV_ABL_SWEEP_TILE_ROWS) and launches them with the row index aliased into
16 384 LDS rows (variant ablation 5 / 6): y is WRONG by design, the loads, the
gathers, the LDS adds and the y stores are those of the real thing.

    python tools/tall_tile_probe.py [--rows 10000000] > profiles/r06_tall_tile_probe.md
"""
import argparse
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if not os.environ.get("SPMV_LIB"):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "spmv_scpa_amd",
                                                     "csrc"), "abl"], check=True)
    os.environ["SPMV_LIB"] = os.path.join(ROOT, "spmv_scpa_amd", "lib",
                                          "libspmv_scpa_amd_abl.so")
import spmv_scpa_amd as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--cols", type=int, default=0)
    ap.add_argument("--tiles", default="0,19552,26080,39104,52096,71584")
    a = ap.parse_args()
    M = a.rows
    N = a.cols or M
    assert S.build_flavour() == "ablations", S.LIB_PATH
    d_x, d_y = S.DevBuffer(N * 8), S.DevBuffer(M * 8)
    S.dev_fill_synth(d_x.ptr, N, 7)
    dA = S.CsrDevice.generate(S.SYNTH_RANDOM, M, N, 32, 2 * N, 0, 42)
    dH = dA.to_hll(True)
    dA.release()
    print("# tall-tile probe, random %d x %d, 32 per row, columns anywhere" %
          (M, N))
    print("# library: %s (%s)" % (S.LIB_PATH, S.build_flavour()))
    print("| tile rows | layout | lambda (entries per x line per bucket) | "
          "launch | ms (median of 20) | note |")
    print("|---|---|---|---|---|---|")
    for tr in [int(v) for v in a.tiles.split(",")]:
        if tr:
            os.environ["SPMV_ABL_SWEEP_TILE_ROWS"] = str(tr)
        else:
            os.environ.pop("SPMV_ABL_SWEEP_TILE_ROWS", None)
        dH.build_panels(0, "sweep")
        desc = dH.panels_describe()
        trows = dH.panels_tile_rows()
        import re
        m = re.search(r"panels=(\d+) x 2\^(\d+)", desc)
        panels, shift = int(m.group(1)), int(m.group(2))
        lam = trows * 32.0 / panels / ((1 << shift) / 16.0)
        for waves, groups, label in ((8, 2, "512 x 2"), (16, 2, "1024 x 2"),
                                     (8, 1, "512 x 1"), (16, 1, "1024 x 1"),
                                     (4, 2, "256 x 2")):
            if tr:  # ablation 5: two groups of 4 per lane, 6: one
                variant = (5 if groups == 2 else 6) << 8
            else:   # the product launch of that shape (bit 11 flips groups)
                variant = 2048 if (waves > 8) == (groups == 2) else 0
                if waves < 8:
                    variant = 0
            try:
                ms = float(np.median(dH.time(
                    S.HLL_KERNEL_PANELS, d_x.ptr, d_y.ptr, 3, 20,
                    waves_per_block=waves, variant=variant)))
            except OSError as e:
                print("| %d | %s | %.2f | %s | - | %s |" % (trows, desc, lam,
                                                              label, e))
                continue
            print("| %d | %s | %.2f | %s | %.4f | %s |" % (
                trows, desc, lam, label, ms,
                "product kernel, y right" if tr == 0 else
                "aliased LDS index (y wrong by design)"))
            sys.stdout.flush()
    dH.release()


if __name__ == "__main__":
    main()

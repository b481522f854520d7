#!/usr/bin/env python3
"""Kernel sweep on one GPU: families x windows x kernels x waves -> ms, GB/s.

    python tools/sweep.py [--rows 10000000] [--k 32] [--cases hll,csr] \
        [--windows 2048,16384,1048576,0] [--waves 4,8] [--iters 10]

W=0 means "columns anywhere".  Prints one line per measurement and a JSON
dump at the end (gpurun_out/sweep.json when run on the GPU box)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def want_ablations(argv):
    """experiment bits of --variants (anything beyond the documented order /
    load bits) exist only in the -DSPMV_ABLATIONS flavour of the library: the
    sweep builds it on demand (`make abl`) and loads it through SPMV_LIB"""
    public = 1 | 2 | 4 | 16 | 32 | 64 | 512 | (1 << 29)
    try:
        vs = argv[argv.index("--variants") + 1]
    except (ValueError, IndexError):
        return False
    return any(int(v) & ~public for v in vs.split(",") if v.strip())


if want_ablations(sys.argv) and not os.environ.get("SPMV_LIB"):
    import subprocess
    abl = os.path.join(ROOT, "spmv_scpa_amd", "lib", "libspmv_scpa_amd_abl.so")
    # always through make: a no-op when the flavour is up to date, a rebuild
    # when a kernel source is newer than it (a stale flavour times old code)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "spmv_scpa_amd",
                                                     "csrc"), "abl"], check=True)
    os.environ["SPMV_LIB"] = abl
import spmv_scpa_amd as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--cols", type=int, default=0,
                    help="columns (default = rows); > rows emulates one "
                         "rank's shard of a multi-GPU problem")
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--family", default="random")
    ap.add_argument("--windows", default="2048,16384,1048576,0")
    ap.add_argument("--hll-kernels", default="1,2")
    ap.add_argument("--csr-kernels", default="2,4")
    ap.add_argument("--waves", default="4")
    ap.add_argument("--groups", default="0")
    ap.add_argument("--variants", default="0")
    ap.add_argument("--panel-cols", default="0")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--flush", type=int, default=0)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    print("# library: %s (%s)" % (S.LIB_PATH, S.build_flavour()))
    kind = {"banded": 0, "random": 1, "ragged": 2, "kkt": 3,
            "stencil": 4, "powerlaw": 5, "hub": 6}[a.family]
    M = a.rows
    N = a.cols or a.rows
    d_x = S.DevBuffer(N * 8)
    d_y = S.DevBuffer(M * 8)
    S.dev_fill_synth(d_x.ptr, N, 7)
    res = []
    ints = lambda s: [int(v) for v in s.split(",") if v != ""]
    for W in ints(a.windows):
        Weff = W if (W > 0 or kind == 4) else 2 * N
        dA = S.CsrDevice.generate(kind, M, N, a.k, Weff, 0, 42)
        mats = {}
        for k in ints(a.hll_kernels):
            cm = True if k == 4 else S.HLL_KERNEL_COL_MAJOR[k]
            if cm not in mats:
                mats[cm] = dA.to_hll(cm)
        runs = [("hll", k, mats[True if k == 4 else S.HLL_KERNEL_COL_MAJOR[k]],
                 pc if k == 4 else 0)
                for k in ints(a.hll_kernels)
                for pc in (ints(a.panel_cols) if k == 4 else [0])]
        runs += [("csr", k, dA, g) for k in ints(a.csr_kernels)
                 for g in (ints(a.groups) if k == 2 else
                           ints(a.panel_cols) if k == 5 else [0])]
        for fmt, k, m, g in runs:
            if (fmt, k) in (("hll", 4), ("csr", 5)):
                import time as _t
                t0 = _t.time()
                m.build_panels(g)
                S.stream_sync()
                print("   panels built in %.2f s (panel_cols=%d)"
                      % (_t.time() - t0, g), flush=True)
            for w, v in [(w, v) for w in ints(a.waves) for v in ints(a.variants)]:
                kw = dict(warmup=2, iters=a.iters, flush_bytes=a.flush,
                          waves_per_block=w, variant=v)
                if fmt == "csr" and k == 2:
                    kw["group"] = g
                ms = float(np.median(m.time(k, d_x.ptr, d_y.ptr, **kw)))
                b = m.kernel_bytes(k)
                r = dict(family=a.family, W=W, fmt=fmt, kernel=k, waves=w,
                         group=g, variant=v, ms=round(ms, 4),
                         gflops=round(2 * m.NZ / ms / 1e6, 1),
                         gbps=round(b / ms / 1e6, 1),
                         frac=round(b / ms / 1e6 / 8000, 4))
                res.append(r)
                print("%-7s W=%-9d %s k%d w%-2d g%-2d v%d  %8.4f ms  %7.1f GF/s  "
                      "%7.1f GB/s  %5.1f%%" % (a.family, W, fmt, k, w, g, v, ms,
                                               r["gflops"], r["gbps"],
                                               100 * r["frac"]), flush=True)
        for m in mats.values():
            m.release()
        dA.release()
    if a.out:
        json.dump(res, open(a.out, "w"), indent=0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""What the measured selector picks, per workload: family x window ->
kernel, layout of the blocked copy, event-timed ms, GFLOP/s, % of 8 TB/s.

    python tools/autotune_report.py [--rows 10000000] [--out profiles/rNN_autotune_report.txt]

One line per workload (HLL col-major and CSR handles of the same matrix);
working sets under 512 MB are timed flushed (1 GiB read-only sweep), like
bench.py does."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import spmv_scpa_amd as S  # noqa: E402

CASES = [  # (label, family, rows, K, window; 0 = columns anywhere)
    ("config2 banded 1M x 16", "banded", 1_000_000, 16, 0),
    ("banded 10M x 32", "banded", 10_000_000, 32, 0),
    ("random W=2^11", "random", 10_000_000, 32, 1 << 11),
    ("random W=2^14", "random", 10_000_000, 32, 1 << 14),
    ("random W=2^17", "random", 10_000_000, 32, 1 << 17),
    ("random W=2^20", "random", 10_000_000, 32, 1 << 20),
    ("random W=2^22", "random", 10_000_000, 32, 1 << 22),
    ("config3 random W=N", "random", 10_000_000, 32, 0),
    ("ragged 24..40 W=2^14", "ragged", 10_000_000, 32, 1 << 14),
    ("stencil 27-point 203^3", "stencil", 203 ** 3, 27, 0),
    ("kkt skewed rows 8.3M", "kkt", 8_345_600, 16, 1 << 16),
]
# the reference's irregular classes (scripts/download-matrices.py:7-38):
# webbase-1M / amazon0302 / roadNet-PA (mean 2-5 entries per row, power-law
# tail) and dc1 (one row and one column far heavier than the rest)
IRREGULAR = [
    ("powerlaw K=3 1M anywhere", "powerlaw", 1_000_000, 3, 0),
    ("powerlaw K=3 4M anywhere", "powerlaw", 4_000_000, 3, 0),
    ("powerlaw K=3 4M W=4096", "powerlaw", 4_000_000, 3, 4096),
    ("powerlaw K=8 2M anywhere", "powerlaw", 2_000_000, 8, 0),
    ("hub K=6 1M W=4096", "hub", 1_000_000, 6, 4096),
    ("hub K=3 4M anywhere", "hub", 4_000_000, 3, 0),
    ("hub K=6 117K W=512 (dc1 size)", "hub", 116_835, 6, 512),
]
KIND = {"banded": 0, "random": 1, "ragged": 2, "kkt": 3, "stencil": 4,
        "powerlaw": 5, "hub": 6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--only", default="")
    ap.add_argument("--rows", type=int, default=0,
                    help="replace the 10M-row cases by this many rows")
    ap.add_argument("--logs", action="store_true",
                    help="print the selector's phase log for every row")
    ap.add_argument("--irregular", action="store_true",
                    help="the power-law / hub families instead of the default list")
    a = ap.parse_args()
    cases = IRREGULAR if a.irregular else CASES
    if a.rows:
        cases = [(lab.replace("10M", "%gM" % (a.rows / 1e6)), fam, a.rows, K, W)
                 for lab, fam, M, K, W in CASES if M == 10_000_000]
    lines = ["# spmv_*_autotune picks, one MI355X (%s), %s"
             % (S.device_info(0)[0], time.strftime("%Y-%m-%d")),
             "# workload | format | pick | ms | GFLOP/s | %% of 8 TB/s | "
             "tune s | layout of the blocked copy",
             "#   + per-kernel medians of the selector (ms; '-' = not a "
             "candidate for this matrix); HLL: stored slots / nnz",
             "#   + the selector's phase log when it took more than 1 s"]
    for label, fam, M, K, W in cases:
        if a.only and a.only not in label:
            continue
        N = M
        Weff = W if (W > 0 or fam == "stencil") else 2 * N
        d_x, d_y = S.DevBuffer(N * 8), S.DevBuffer(M * 8)
        S.dev_fill_synth(d_x.ptr, N, 7)
        dA = S.CsrDevice.generate(KIND[fam], M, N, K, Weff, 0, 42)
        dH = dA.to_hll(True)
        for fmt, m, labels, pid in (("HLL", dH, S.HLL_KERNEL_LABELS,
                                     S.HLL_KERNEL_PANELS),
                                    ("CSR", dA, S.CSR_KERNEL_LABELS,
                                     S.CSR_KERNEL_PANELS)):
            t0 = time.time()
            best, _ = m.autotune(d_x.ptr, d_y.ptr)
            tune = time.time() - t0
            flush = (1 << 30) if m.algorithmic_bytes < (512 << 20) else 0
            ms = float(np.median(m.time(best, d_x.ptr, d_y.ptr, 3, 20, flush)))
            b = m.kernel_bytes(best)  # blocked copy of an HLL handle: true entries
            lines.append("%-24s | %s | %-22s | %8.4f | %7.1f | %5.1f | %4.1f | %s"
                         % (label, fmt, labels[best], ms,
                            2 * m.NZ / ms / 1e6, 100 * b / ms / 1e6 / 8000,
                            tune, m.panels_describe() if best == pid else "-"))
            print(lines[-1], flush=True)
            tm = m.tune_times()
            detail = "    selector ms: " + "  ".join(
                "%s %s" % (labels[k], ("%.4f" % tm[k]) if tm[k] > 0 else "-")
                for k in range(len(tm)))
            if fmt == "HLL":
                detail += "   | slots / nnz = %.2f" % (m.slots / max(m.NZ, 1))
            else:
                detail += "   | rows %d, nnz %d, mean %.2f" % (
                    m.M, m.NZ, m.NZ / max(m.M, 1))
            lines.append(detail)
            print(detail, flush=True)
            if tune > 1.0 or a.logs:
                for ln in (m.tune_log() or "").splitlines():
                    lines.append("    log: " + ln)
                    print(lines[-1], flush=True)
        dH.release()
        dA.release()
        d_x.free()
        d_y.free()
    if a.out:
        open(a.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()

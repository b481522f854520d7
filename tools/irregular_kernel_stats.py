#!/usr/bin/env python3
"""profiles/<round>_irregular_kernel_stats.md from the rocprofv3
--kernel-trace --stats runs of `tools/autotune_report.py --irregular --only
<case>` (gpurun_out/prof_<round>_irr_<case>/): per kernel calls, average /
min / max duration -- the side launches of the long rows and wide hack blocks
next to the kernels they complete, and the selector's build kernels.

    python tools/irregular_kernel_stats.py r04
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = (("hub K=6 1M W=4096 (a row of 131072 entries + a hub column)",
          "hub_K_6_1M"),
         ("powerlaw K=3 4M, columns anywhere", "powerlaw_K_3_4M_anywhere"))
KEEP = ("k_csr_", "k_hll_", "k_tiles_", "k_long_rows", "k_long_copy", "k_keys",
        "k_tile_gather")


def main():
    rnd = sys.argv[1]
    out = ["# rocprofv3 --kernel-trace --stats of `tools/autotune_report.py "
           "--irregular --only <case>`",
           "# (the selector + 20 timed launches of its pick per handle), one "
           "MI355X, round %s build." % rnd[1:].lstrip("0"),
           "# Per kernel: calls, average / min / max duration in us.  "
           "`k_long_rows`, `k_csr_long_seg`, `k_hll_wide`: the side launches",
           "# of the long rows / wide hack blocks, next to the kernels they "
           "complete.", ""]
    for title, tag in CASES:
        fns = sorted(glob.glob(os.path.join(
            ROOT, "gpurun_out", "prof_%s_irr_%s" % (rnd, tag), "**",
            "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
        if not fns:
            raise SystemExit("no kernel_stats.csv for " + tag)
        rows = list(csv.DictReader(open(fns[-1])))
        out += ["## " + title, "", "| kernel | calls | avg us | min us | max us |",
                "|---|---|---|---|---|"]
        for r in sorted((r for r in rows if any(k in r["Name"] for k in KEEP)),
                        key=lambda r: -float(r["TotalDurationNs"])):
            out.append("| `%s` | %s | %.1f | %.1f | %.1f |" % (
                r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3,
                float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
        out.append("")
    dst = os.path.join(ROOT, "profiles", "%s_irregular_kernel_stats.md" % rnd)
    open(dst, "w").write("\n".join(out))
    print("\n".join(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Batch runner: every .mtx of a directory (and/or synthetic families) through
the spmv_scpa_amd driver, N iterations each, CSVs appended in one directory.

Does the job of the reference's scripts/results.py (same -m / -res / -i
meaning; -exe defaults to this repo's driver) and prints, at the end, the
per-(matrix, format, kernel, waves) medians the reference's scripts/plots.py
computes before plotting (plots.py:21-53), so the CSVs can be checked without
pandas/matplotlib.  serial.csv / omp.csv / cuda.csv keep the reference's
columns, so plots.py reads them unchanged.

    python tools/run_all.py -m matrices/ -res results/ -i 5
    python tools/run_all.py -res results/ -i 3 --synthetic random:1000000:32:65536
"""
import argparse
import collections
import csv
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "spmv_scpa_amd", "bin", "spmv_scpa_amd")


def medians(path, keys):
    rows = collections.defaultdict(list)
    if not os.path.isfile(path):
        return {}
    for r in csv.DictReader(open(path)):
        rows[tuple(r[k] for k in keys)].append(
            (float(r["duration_ms"]), float(r["gflops"])))
    return {k: (statistics.median(v[0] for v in vs),
                statistics.median(v[1] for v in vs), len(vs))
            for k, vs in rows.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-exe", default=DRIVER)
    ap.add_argument("-m", default=None, help="directory of .mtx files")
    ap.add_argument("-res", required=True, help="directory for the CSV files")
    ap.add_argument("-i", type=int, default=10, help="iterations per matrix")
    ap.add_argument("--synthetic", action="append", default=[],
                    help="family:rows:nnz_per_row:window (repeatable)")
    ap.add_argument("--debug", action="store_true", help="pass -d (validate)")
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    os.makedirs(a.res, exist_ok=True)
    jobs = []
    if a.m:
        for f in sorted(os.listdir(a.m)):
            if f.endswith(".mtx") and os.path.isfile(os.path.join(a.m, f)):
                jobs.append((f[:-4], ["-m", os.path.join(a.m, f)]))
    for spec in a.synthetic:
        fam, rows, k, w = (spec.split(":") + ["0"] * 3)[:4]
        jobs.append(("synth_" + fam, ["-s", fam, "--rows", rows, "--nnz-row", k,
                                      "--window", w]))
    if not jobs:
        print("[ERROR] nothing to run: give -m <dir> and/or --synthetic")
        return 1
    extra = (["-d"] if a.debug else []) + (["--no-cpu"] if a.no_cpu else [])
    for name, args in jobs:
        for it in range(a.i):
            print("[%s] iteration %d/%d" % (name, it + 1, a.i), flush=True)
            r = subprocess.run([a.exe] + args + ["-o", a.res] + extra)
            if r.returncode:
                print("[ERROR] %s failed at iteration %d (rc %d)"
                      % (name, it + 1, r.returncode))
    gpu = medians(os.path.join(a.res, "cuda.csv"),
                  ["matrix", "format", "kernel", "warps_per_block"])
    print("\nGPU medians (matrix, format, kernel, waves): ms, GFLOP/s, runs")
    for k in sorted(gpu):
        print("  %-40s %10.4f %10.2f %4d" % (" ".join(k), *gpu[k]))
    ser = medians(os.path.join(a.res, "serial.csv"), ["matrix", "format"])
    print("serial medians (matrix, format): ms, GFLOP/s, runs")
    for k in sorted(ser):
        print("  %-40s %10.4f %10.2f %4d" % (" ".join(k), *ser[k]))
    return 0


if __name__ == "__main__":
    sys.exit(main())

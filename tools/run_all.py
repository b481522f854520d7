#!/usr/bin/env python3
"""Batch runner: every .mtx of a directory (and/or synthetic families) through
the spmv_scpa_amd driver, N iterations each, over a list of GPU counts, CSVs
appended in one directory.

Does the job of the reference's scripts/results.py (same -m / -res / -i
meaning, results.py:17-28; -exe defaults to this repo's driver) extended by
the GPU-count sweep of SURVEY 8f-4, and prints, at the end, the
per-(matrix, format, kernel, waves) medians the reference's scripts/plots.py
computes before plotting (plots.py:21-53), so the CSVs can be checked without
pandas/matplotlib.  serial.csv / omp.csv / cuda.csv keep the reference's
columns, so plots.py reads them unchanged; multi-GPU steps (kernels +
all-gather of y) go to roofline.csv with their `gpus` column.

    python tools/run_all.py -m matrices/ -res results/ -i 5
    python tools/run_all.py -m matrices/ -res results/ -i 5 --gpus 1,2,4,8
    python tools/run_all.py -res results/ -i 3 --synthetic random:1000000:32:65536

For every matrix the first GPU count runs the whole grid (CPU benchmarks,
every single-GPU kernel x waves, then `-g N`); the further counts run only
the `-g N` step (`--only-multi-gpu`).  Counts above the number of visible
GPUs are skipped with a note.
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

# the reference's CSV schemas (logger.c:19-54; read by plots.py:21-53)
SERIAL_COLS = ["matrix", "format", "rows", "cols", "nnz", "num_blocks",
               "duration_ms", "gflops"]
OMP_COLS = ["matrix", "format", "bench", "rows", "cols", "nnz", "num_blocks",
            "num_threads", "duration_ms", "gflops"]
CUDA_COLS = ["matrix", "format", "kernel", "warps_per_block", "rows", "cols",
             "nnz", "num_blocks", "duration_ms", "gflops"]


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


def visible_gpus():
    sys.path.insert(0, ROOT)
    try:
        import spmv_scpa_amd as S
        return S.device_count()
    except Exception:  # noqa: BLE001 - the library is optional for -exe runs
        return 0


def plan(jobs, gpu_counts, iterations, extra):
    """the driver invocations, in order: [(label, argv-after-exe), ...]"""
    out = []
    for name, args in jobs:
        for it in range(iterations):
            for n, g in enumerate(gpu_counts or [None]):
                argv = list(args) + extra
                if g is not None:
                    argv += ["-g", str(g)]
                    if n > 0:
                        argv.append("--only-multi-gpu")
                out.append(("[%s] iteration %d/%d%s"
                            % (name, it + 1, iterations,
                               "" if g is None else ", %d GPU(s)" % g), argv))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-exe", default=DRIVER)
    ap.add_argument("-m", default=None, help="directory of .mtx files")
    ap.add_argument("-res", required=True, help="directory for the CSV files")
    ap.add_argument("-i", type=int, default=10, help="iterations per matrix")
    ap.add_argument("--gpus", default="",
                    help="comma-separated GPU counts, e.g. 1,2,4,8: each "
                         "matrix also runs row-partitioned over that many "
                         "GPUs (driver flag -g)")
    ap.add_argument("--partition", default="even", choices=["even", "nnz"],
                    help="--gpus: row ranges of equal row counts, or of "
                         "near-equal entry counts (driver flag --partition; "
                         "the multi-GPU form of the reference's "
                         "partition_csr_rows, csr.c:218-276)")
    ap.add_argument("--ragged-exchange", default="p2p",
                    choices=["p2p", "bcast", "padded"],
                    help="--partition nnz: how the ragged y fragments travel")
    ap.add_argument("--assume-gpus", type=int, default=-1,
                    help="do not query the device count (tests, remote exe)")
    ap.add_argument("--synthetic", action="append", default=[],
                    help="family:rows:nnz_per_row:window (repeatable)")
    ap.add_argument("--debug", action="store_true", help="pass -d (validate)")
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args(argv)
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
    counts = sorted({int(v) for v in a.gpus.split(",") if v.strip()})
    if any(g < 1 for g in counts):
        print("[ERROR] --gpus wants positive counts")
        return 1
    if counts:
        have = a.assume_gpus if a.assume_gpus >= 0 else visible_gpus()
        skipped = [g for g in counts if g > have]
        counts = [g for g in counts if g <= have]
        if skipped:
            print("[NOTE] %d GPU(s) visible: skipping the counts %s"
                  % (have, ",".join(map(str, skipped))))
    extra = ["-o", a.res] + (["-d"] if a.debug else []) + \
        (["--no-cpu"] if a.no_cpu else []) + \
        (["--partition", a.partition, "--ragged-exchange", a.ragged_exchange]
         if a.partition != "even" else [])
    failed = 0
    for label, argv_ in plan(jobs, counts, a.i, extra):
        print(label, flush=True)
        r = subprocess.run([a.exe] + argv_)
        if r.returncode:
            failed += 1
            print("[ERROR] %s failed (rc %d)" % (label, r.returncode))
    gpu = medians(os.path.join(a.res, "cuda.csv"),
                  ["matrix", "format", "kernel", "warps_per_block"])
    print("\nGPU medians (matrix, format, kernel, waves): ms, GFLOP/s, runs")
    for k in sorted(gpu):
        print("  %-40s %10.4f %10.2f %4d" % (" ".join(k), *gpu[k]))
    ser = medians(os.path.join(a.res, "serial.csv"), ["matrix", "format"])
    print("serial medians (matrix, format): ms, GFLOP/s, runs")
    for k in sorted(ser):
        print("  %-40s %10.4f %10.2f %4d" % (" ".join(k), *ser[k]))
    omp = medians(os.path.join(a.res, "omp.csv"),
                  ["matrix", "format", "bench", "num_threads"])
    print("OpenMP medians (matrix, format, bench, threads): ms, GFLOP/s, runs")
    for k in sorted(omp, key=lambda t: (t[0], t[1], t[2], int(t[3]))):
        print("  %-40s %10.4f %10.2f %4d" % (" ".join(k), *omp[k]))
    roof = medians(os.path.join(a.res, "roofline.csv"),
                   ["matrix", "format", "kernel", "gpus"])
    if roof:
        print("steps incl. all-gather (matrix, format, kernel, gpus): ms, "
              "GFLOP/s, runs")
        for k in sorted(roof, key=lambda t: (t[0], t[1], int(t[3]), t[2])):
            print("  %-40s %10.4f %10.2f %4d" % (" ".join(k), *roof[k]))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())

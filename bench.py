#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X SpMV engine.

Metric (BASELINE.json): fp64 SpMV GFLOP/s + achieved HBM GB/s (% of the
8 TB/s roofline) at 1/2/4/8 MI355X.  Default workload (BASELINE.json
configs[2], the one the >=60 %-of-roofline target is quoted on; configs[4]
at 8 GPUs): synthetic random HLL, hack 32, 32 nnz/row, 10M rows PER GPU (weak
scaling), N = 10M x n_gpus columns ANYWHERE (W = N, the family's worst case),
rows partitioned by contiguous ranges, x replicated, every rank computes its
y fragment and the fragments are all-gathered in place over RCCL
(torch.distributed backend "nccl").  SURVEY 8d asks for W = N and the banded
members of the family side by side: `roofline.variants` carries W = 2^20 and
W = 2^17 measured in the same run.

A step = one SpMV over the whole matrix (+ the all-gather of y when N > 1),
inputs resident in HBM.  value = 2 * nnz_global / step time, in GFLOP/s
(reference definition, include/utils.h:70-75).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N
    python bench.py --config 4     # nlpkkt160-sized .mtx through the loader
    python bench.py --config 2     # 1M banded CSR, 16/row, flushed

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def host_cpus():
    """(threads visible to this process, cgroup CPU quota or None)"""
    try:
        vis = len(os.sched_getaffinity(0))
    except AttributeError:
        vis = os.cpu_count() or 1
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(p)
    except (OSError, ValueError):
        pass
    return vis, quota


# The bench process itself needs almost no host threads, but libgomp (numpy,
# torch and the product library's host generator all share it) sizes its team
# from the affinity mask -- 256 on the GPU box -- not from the cgroup quota
# (16 CPUs there).  Round 2's driver line lost 4.4 ms per step to exactly
# that: full-width teams spinning after tiny parallel regions burned the CFS
# quota and the launching thread was throttled inside the timed loop.  So,
# BEFORE anything loads libgomp: team size <= quota, idle workers sleep.
# (The cpu_baseline child gets the ORIGINAL environment back, _ENV0.)
_ENV0 = dict(os.environ)


def cap_openmp_env():
    """first thing main() does; returns the team size it set (or found)"""
    vis, quota = host_cpus()
    os.environ.setdefault(
        "OMP_NUM_THREADS",
        str(max(1, min(vis, int(quota)) if quota else vis)))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    return int(os.environ["OMP_NUM_THREADS"])


def cgroup_cpu_stat():
    """nr_periods / nr_throttled / throttled_usec of this cgroup (v2), {} when
    not readable: evidence for or against CFS throttling of the host thread"""
    out = {}
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, v = line.split()
            if k in ("nr_periods", "nr_throttled", "throttled_usec"):
                out[k] = int(v)
    except (OSError, ValueError):
        pass
    return out


def stat_delta(a, b):
    return {k: b[k] - a[k] for k in b if k in a}

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# Working sets under 512 MB are pushed out of the 256 MiB Infinity Cache
# between timed launches by a READ-ONLY sweep of a 1 GiB scratch
# buffer: four times the cache (a 512 MiB sweep left the stream kernel 3.5 %
# faster, i.e. not everything was gone), and no dirty lines whose write-back
# would overlap the timed launch (engine.hip k_flush_ro; A/B in DESIGN.md).
FLUSH_BYTES = 1 << 30
ROWS_PER_GPU = 10_000_000
NNZ_PER_ROW = 32
MATRIX_SEED, X_SEED = 42, 7
FAMILIES = {"banded": 0, "random": 1, "ragged": 2, "kkt": 3, "stencil": 4,
            "powerlaw": 5, "hub": 6}
# the reference's thread ladder (src/main.c:176-180) + serial + all cores
REF_LADDER = (2, 4, 8, 16, 32, 40)
METRIC = ("fp64 SpMV GFLOP/s + achieved HBM GB/s (% of roofline), "
          "1/2/4/8 MI355X")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 4],
                    help="BASELINE.json config: 3 = 10M x 10M random HLL "
                         "(headline, default), 2 = 1M banded CSR (flushed), "
                         "4 = nlpkkt160-sized .mtx through the loader (CSR)")
    ap.add_argument("--rows-per-gpu", type=int, default=ROWS_PER_GPU)
    ap.add_argument("--nnz-row", type=int, default=NNZ_PER_ROW)
    ap.add_argument("--window", type=int, default=0,
                    help="column window W of the random family; 0 = N "
                         "(columns anywhere: the worst case)")
    ap.add_argument("--family", default="random", choices=sorted(FAMILIES))
    ap.add_argument("--format", default="hll", choices=["hll", "csr"])
    ap.add_argument("--kernel", type=int, default=-1,
                    help="kernel id (hip_hll.h / hip_csr.h); -1 = autotuned")
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--blocked-pin", default="",
                    help="run the 2-D blocked kernel on exactly this layout "
                         "(the `config.blocked_pin` string of an earlier "
                         "line) instead of asking the selector: the counter "
                         "passes of tools/profile.sh measure the layout the "
                         "un-profiled run picked")
    ap.add_argument("--chunks", type=int, default=0,
                    help="N>1: split each shard into row chunks and overlap "
                         "the all-gather of chunk c with the kernel of c+1 "
                         "(0 = auto: 4 when N > 1, else 1)")
    ap.add_argument("--shards-per-gpu", type=int, default=1,
                    help="logical shards of --rows-per-gpu rows held by each "
                         "GPU (each its own int32-safe matrix)")
    ap.add_argument("--strong", action="store_true",
                    help="BASELINE config 5 as a FIXED problem: 8 logical "
                         "shards of --rows-per-gpu rows (80M x 80M), 8/N per "
                         "GPU; strong scaling over N = 1, 2, 4, 8")
    ap.add_argument("--no-strong-leg", action="store_true",
                    help="N>1: skip the extra fixed-problem measurement "
                         "reported in config.strong")
    ap.add_argument("--reserve-cus", type=int, default=8,
                    help="N>1, sweep schedule: compute units left to RCCL's "
                         "kernels in the overlapped arrangement")
    ap.add_argument("--exchange", default="auto", choices=["auto", "halo"],
                    help="halo: only the rows within --halo-rows of another "
                         "rank's range travel (opt-in; NOT the all-gather "
                         "path BASELINE names)")
    ap.add_argument("--halo-rows", type=int, default=0)
    ap.add_argument("--force-exchange", action="store_true",
                    help="initialise RCCL and run the y exchange even with "
                         "one rank (exercises the multi-GPU path on a "
                         "1-GPU box)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend.  nccl (= RCCL) is the "
                         "product path.  gloo is a REHEARSAL of the multi-rank "
                         "control flow on a box with fewer GPUs than ranks: "
                         "ranks share the visible GPUs (rank %% device count) "
                         "and y fragments are staged through host memory -- "
                         "its timings mean nothing")
    ap.add_argument("--mtx", default="",
                    help="--config 4: Matrix Market file (default: "
                         "$SPMV_MTX_DIR/nlpkkt160.mtx, else the generated "
                         "nlpkkt160-shaped file)")
    ap.add_argument("--kkt-n", type=int, default=160,
                    help="--config 4: grid edge of the generated file")
    ap.add_argument("--cpu-csv-dir", default="",
                    help="where the cpu_baseline leg appends serial.csv / "
                         "omp.csv rows (reference schema); default "
                         "gpurun_out/cpu_baseline")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launcher self-test: every rank joins the process "
                         "group, one all-reduce, rank 0 prints a line with "
                         "n_gpus and no measurement (runs without a GPU on "
                         "the gloo backend)")
    ap.add_argument("--native-mgpu", action="store_true",
                    help="measure the product library's OWN multi-GPU entry "
                         "points instead of torch.distributed: ONE process, "
                         "spmv_mgpu_* (include/spmv_mgpu.h; mgpu.hip: "
                         "ncclCommInitAll + grouped in-place ncclAllGather), "
                         "same workload, same JSON shape")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    return ap.parse_args(argv)


# ---------------------------------------------------------------- CPU baseline
def thread_ladder(nproc, quota=None):
    """serial is always timed; OpenMP at the reference's counts (src/main.c:
    176-180) and at "all cores": every visible hardware thread, or -- under a
    cgroup CPU quota smaller than that -- the quota, which is all the cores
    this process can actually run on (256 threads on 16 CPUs of quota measured
    1.3 GFLOP/s in round 3: an oversubscription figure, not a baseline)"""
    allc = nproc if not quota or quota >= nproc else max(1, int(quota + 0.999))
    return sorted({t for t in REF_LADDER if t <= nproc} | {allc})


def log_cpu_rows(S, out_dir, name, M, N, nnz, runs, hll_blocks=0):
    """append the runs to serial.csv / omp.csv through the product's logger
    (reference schema, logger.c:19-54)"""
    import ctypes as C
    os.makedirs(out_dir, exist_ok=True)
    if S._lib.logger_init(os.fsencode(out_dir)) != 0:
        return None
    hdr = S.SparseCSR()
    hdr.name = name.encode()[:63]
    hdr.M, hdr.N, hdr.NZ = M, N, nnz
    hh = S.SparseHLL()
    hh.name = name.encode()[:63]
    hh.M, hh.N, hh.NZ = M, N, nnz
    hh.hack_size, hh.num_blocks = S.HACK_SIZE, hll_blocks
    for r in runs:
        b = S.Bench()
        b.duration_ms, b.gflops = r["median_ms"], r["gflops"]
        hll = r["format"] == "HLL"
        if r["bench"] == "serial":
            if hll:
                S._lib.log_hll_serial_benchmark(C.byref(hh), b)
            else:
                S._lib.log_csr_serial_benchmark(C.byref(hdr), b)
        else:
            bo = S.BenchOmp()
            bo.name = r["bench"].encode()
            bo.bench, bo.num_threads = b, r["threads"]
            if hll:
                S._lib.log_hll_omp_benchmark(C.byref(hh), bo)
            else:
                S._lib.log_csr_omp_benchmark(C.byref(hdr), bo)
    S._lib.logger_close()
    return out_dir


CPU_WINDOW_MS = 300  # >= 3 CFS periods of 100 ms per sample


def cpu_baseline(S, kind, M, N, K, W, csv_dir, name, reps=3):
    """The reference's own serial + OpenMP path (oracle/_ref/ref_fast, built
    from /root/reference/src by oracle/build_ref.sh with the reference's
    flags) on the SAME full-size input: CSR at the thread ladder {1, 2, 4, 8,
    16, 32, 40, all cores} (src/main.c:176-180), then the HLL legs
    (hll.c:127-150, 178-211) after ONE csr_to_hll: serial and OpenMP at the
    thread count that was best for CSR.  EVERY leg: median of `reps` (>= 3)
    samples, a sample = the reference's single-shot bench repeated until
    CPU_WINDOW_MS of run time is covered -- round 3's driver line carried a
    one-shot 44 ms HLL run at 40 threads under a 16-CPU quota (14.4 GFLOP/s;
    6.3 on another box): shorter than one CFS period, it ran on burst credit.
    `value` = the best median, CSR or HLL.  OMP_PROC_BIND=close.  Falls back
    to the oracle port."""
    nproc, quota = host_cpus()
    ladder = thread_ladder(nproc, quota)
    qtxt = ("cgroup quota %g CPUs of %d visible hardware threads" %
            (quota, nproc)) if quota else "%d hardware threads" % nproc
    sample = ("full size: %s %dx%d, %d nnz/row, W=%s, same generator and "
              "seeds as the GPU run; CSR serial + omp_guided + omp_nnz at "
              "threads %s, then HLL serial + omp_guided at the best CSR "
              "thread count; every leg: median of %d samples, each the "
              "single-shot bench repeated over >= %d ms; %s"
              % (name, M, N, K, "N" if W >= 2 * N else str(W),
                 "/".join(str(t) for t in ladder), reps, CPU_WINDOW_MS, qtxt))
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_fast")
    env = dict(_ENV0, OMP_NUM_THREADS=str(max(ladder)),
               OMP_PROC_BIND="close", OMP_PLACES="cores",
               REF_TIME_HLL="best", REF_TIME_WINDOW_MS=str(CPU_WINDOW_MS))
    env.pop("OMP_WAIT_POLICY", None)  # the reference runs libgomp's default
    err = "oracle/_ref/ref_fast not present"
    if os.path.exists(ref):
        try:
            t0 = time.time()
            out = subprocess.run(
                [ref, "time", str(kind), str(M), str(N), str(K), str(W),
                 str(MATRIX_SEED), str(X_SEED), str(reps)]
                + [str(t) for t in ladder],
                capture_output=True, text=True, timeout=900, env=env,
                check=True)
            res = json.loads(out.stdout)
            runs = res["runs"]
            best = max(runs, key=lambda r: r["gflops"])
            serial = [r for r in runs if r["bench"] == "serial"
                      and r["format"] == "CSR"][0]
            hll = [r for r in runs if r["format"] == "HLL"]
            logged = log_cpu_rows(S, csv_dir, name, M, N, res["nnz"], runs,
                                  res.get("hll_blocks", 0))
            return {"value": round(best["gflops"], 3), "unit": "GFLOP/s",
                    "cores": best["threads"], "kind": "reference",
                    "sample": sample,
                    "best": "%s %s" % (best["format"], best["bench"]),
                    "serial_csr_gflops": round(serial["gflops"], 3),
                    "best_hll_gflops": round(max(r["gflops"] for r in hll), 3)
                    if hll else None,
                    "host_threads": nproc, "cpu_quota": quota,
                    "reps": reps, "window_ms": CPU_WINDOW_MS,
                    "ladder": [[r["format"], r["bench"], r["threads"],
                                round(r["gflops"], 3)] for r in runs],
                    "hll_convert_s": round(res.get("hll_prep_ms", 0) / 1e3, 1),
                    "csv_dir": logged, "wall_s": round(time.time() - t0, 1)}
        except Exception as e:  # pragma: no cover - depends on the box
            err = "ref_fast failed: %r" % (e,)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    rows = min(M, 2_000_000)  # the port is serial numpy/C glue: keep it short
    IRP, JA, AS = O.synth_csr(kind, rows, N, K, W, MATRIX_SEED)
    x = O.synth_x(X_SEED, 0, N)
    ms1 = O.time_csr_ms(IRP, JA, AS, x, 1, 3)
    msn = O.time_csr_ms(IRP, JA, AS, x, nproc, 3)
    best_ms, thr = (ms1, 1) if ms1 <= msn else (msn, nproc)
    return {"value": round(2.0 * len(JA) / (best_ms * 1e6), 3),
            "unit": "GFLOP/s", "cores": thr, "kind": "port",
            "sample": "first %d rows of: %s" % (rows, sample), "note": err,
            "host_threads": nproc}


# ---- result check without the oracle: rows regenerated by the host-side C
# generator of the product library (include/spmv_synth.h) -- the same
# definition the device generator implements, compiled for the CPU
def host_row_dots(S, kind, N, K, W, seed, xseed, rows):
    """(dots, sum |terms|) of the GLOBAL rows `rows` of the synthetic family
    times x, in ONE serial call of the product library (csr_synth_row_dots:
    no OpenMP team; round 2 regenerated the rows one by one through
    csr_generate / vec_synth, ~8500 parallel regions before the timed loop)"""
    import ctypes as C
    import numpy as np
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    dot = np.zeros(len(rows))
    scale = np.zeros(len(rows))
    fn = S._lib.csr_synth_row_dots
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_uint64,
                   C.c_uint64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rc = fn(kind, N, K, W, seed, xseed, rows.ctypes.data, len(rows),
            dot.ctypes.data, scale.ctypes.data)
    if rc:
        raise OSError(-rc, "csr_synth_row_dots")
    return dot, scale


def host_row_dot(S, kind, N, K, W, seed, xseed, g):
    """(dot, sum |terms|) of global row g (single-row form of the above)"""
    d, sc = host_row_dots(S, kind, N, K, W, seed, xseed, [int(g)])
    return float(d[0]), float(sc[0])


def check_rows(S, kind, N, K, W, got, rows_global):
    """raise SystemExit unless |y - y_host| <= 1e-6 max(|y_host|, 1e-3 sum|a x|)
    on every given row (north star: 1e-6 relative fp64); returns the count"""
    want, scale = host_row_dots(S, kind, N, K, W, MATRIX_SEED, X_SEED,
                                rows_global)
    for g, w, sc, r in zip(got, want, scale, rows_global):
        if abs(g - w) > 1e-6 * max(abs(w), 1e-3 * sc):
            raise SystemExit("parity check failed on row %d: %r vs %r"
                             % (r, g, w))
    if len(want) == 0:
        raise SystemExit("no row of y was checked")
    return len(want)


def kernel_source_blob(kname):
    """blob id of the source file that holds kernel `kname`"""
    import hashlib
    fn = ("panels.hip" if "tile_panels" in kname else
          "hll_kernels.hip" if kname.startswith("hll_") else "csr_kernels.hip")
    try:
        data = open(os.path.join(ROOT, "spmv_scpa_amd", "csrc", fn),
                    "rb").read()
    except OSError:
        return fn, None
    return fn, hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def _git_blob(path):
    import hashlib
    try:
        data = open(path, "rb").read()
    except OSError:
        return None
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def kernel_source_ident(kname):
    """what a profile must agree with to describe this build: the blob ids
    of the kernel's source file AND of hip_common.h (segment sizes, budgets
    and the device helpers the kernels inline live there)"""
    fn, blob = kernel_source_blob(kname)
    return {"file": fn, "blob": blob, "common_file": "hip_common.h",
            "common_blob": _git_blob(os.path.join(ROOT, "spmv_scpa_amd", "csrc",
                                                  "hip_common.h"))}


def same_build(ks, kname):
    """does the `kernel_source` record of a committed profile name the tree's
    sources?  (records without the common header's blob predate the rule)"""
    me = kernel_source_ident(kname)
    ks = ks or {}
    return bool(me["blob"] and me["common_blob"]
                and ks.get("blob") == me["blob"]
                and ks.get("common_blob") == me["common_blob"])


def measured_traffic(workload, kname, schedule=None):
    """-> (traffic dict or None, why-not or None).  HBM-side bytes per launch
    of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*.traffic.json, written by tools/summarize_profile.py from
    `tools/profile.sh` runs of THIS command).  A profile describes this build
    only if it was taken with the same kernel source: the json carries the
    git blob id of the source file (`kernel_source`), and a profile whose
    blob differs from the tree's -- or that predates the field -- is refused,
    so the line can never quote the bytes of another kernel."""
    import glob
    fn_src, blob = kernel_source_blob(kname)
    best, why = None, "no committed profile of this workload + kernel"
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*.traffic.json"))):
        try:
            t = json.load(open(fn))
        except ValueError:
            continue
        if t.get("workload") != workload or t.get("bench_kernel") != kname:
            continue
        if (schedule and t.get("blocked_schedule")
                and t["blocked_schedule"] != schedule):
            why = ("profiles/%s describes the %s schedule of the blocked "
                   "kernel, this run the %s one"
                   % (os.path.basename(fn), t["blocked_schedule"], schedule))
            continue
        ks = t.get("kernel_source") or {}
        if not same_build(ks, kname):
            why = ("profiles/%s was taken with another build of %s / "
                   "hip_common.h (blob %s, tree %s)"
                   % (os.path.basename(fn), fn_src,
                      str(ks.get("blob"))[:12], str(blob)[:12]))
            continue
        best, why = t, None
    return best, why


def workload_name(family, fmt, Mloc, Nglob, Mglob, K, window, W, L=1, Mshard=0):
    wtxt = "column window W=%s" % ("N (anywhere)" if window <= 0 else str(W))
    if family == "banded":
        wtxt = "columns s..s+K-1 around the diagonal"
    elif family == "stencil":
        wtxt = "grid edge %s" % ("cbrt(N)" if window <= 0 else str(W))
    s = ("%s %s %dx%d per GPU (%dx%d global), hack 32, %d nnz/row, %s, seed %d"
         % (family, fmt.upper(), Mloc, Nglob, Mglob, Nglob, K, wtxt,
            MATRIX_SEED))
    if L > 1:
        s += ", %d logical shards of %d rows per GPU" % (L, Mshard)
    return s


def roofline_dict(alg_bytes, kern_ms, kname, nnz, traffic, why=None):
    import numpy as np
    kavg = float(np.mean(kern_ms))
    achieved = alg_bytes / (kavg * 1e6)
    return {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
        "traffic": round(traffic["bytes_per_launch"]) if traffic else None,
        "traffic_source": ("profiles/" + traffic["source"]
                           + " (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE, "
                           "separate passes; same kernel source blob)")
        if traffic else why,
        # the blocked layout the counters were taken on (the selector runs
        # again in every pass: compare with config.blocked_layout)
        "traffic_layout": traffic.get("blocked_layout") if traffic else None,
        "kernel": kname, "algorithmic_bytes_per_launch": alg_bytes,
        "kernel_ms_avg": round(kavg, 5),
        "kernel_ms_min": round(float(np.min(kern_ms)), 5),
        "kernel_gflops": round(2.0 * nnz / (kavg * 1e6), 2),
    }


# The W = N kernel is not HBM-bound: rocprofv3's TCP/TCC counters
# (profiles/r02_tcp_counters_sweep.md, re-collected per round into
# profiles/*.l2req.json by tools/pmc.sh + tools/l2req_profile.py) show every
# CU's vector L1 holding its ~107 outstanding line requests for the whole
# kernel: what the kernel runs out of is L2 line requests in flight.  The
# line therefore carries a second roofline: requests per launch (measured,
# TCP_TCC_READ_REQ summed over the chip) against what the eight L2s accept --
# 16 channels per XCD, one request per channel and clock at 2.4 GHz.
L2_CHANNELS = 128
L2_CLOCK_GHZ = 2.4
TCP_SLOTS = 107  # outstanding line requests a CU's vector L1 tracks (r02)
NUM_CUS = 256


def measured_l2_requests(workload, kname, schedule=None):
    """-> (profile dict or None, why-not): committed *.l2req.json of the same
    workload, kernel and kernel-source blob (same staleness rule as
    measured_traffic)"""
    import glob
    fn_src, blob = kernel_source_blob(kname)
    why = "no committed l2req profile of this workload + kernel"
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*.l2req.json")),
                     reverse=True):
        try:
            t = json.load(open(fn))
        except ValueError:
            continue
        if t.get("workload") != workload or t.get("bench_kernel") != kname:
            continue
        sched = t.get("blocked_schedule") or next(
            (w for w in ("sweep", "chain", "steps")
             if str(t.get("blocked_layout") or "").startswith(w)), None)
        if schedule and sched and sched != schedule:
            why = ("profiles/%s describes the %s schedule, this run the %s one"
                   % (os.path.basename(fn), sched, schedule))
            continue
        if not same_build(t.get("kernel_source"), kname):
            why = ("profiles/%s was taken with another build of %s"
                   % (os.path.basename(fn), fn_src))
            continue
        return t, None
    return None, why


def secondary_roofline(workload, kname, kavg_ms, schedule=None):
    """bound "l2_line_requests" (what the blocked sweep kernel on W = N is
    held by): measured CU->L2 line requests per launch against the L2s' peak
    acceptance rate, and the floor the per-CU outstanding-request capacity
    sets at the measured mean latency (Little's law)"""
    prof, why = measured_l2_requests(workload, kname, schedule)
    if not prof:
        return {"bound": "l2_line_requests", "frac": None, "source": why}
    reqs = float(prof["requests_per_launch"])
    peak = L2_CHANNELS * L2_CLOCK_GHZ * 1e9
    ach = reqs / (kavg_ms * 1e-3)
    out = {"bound": "l2_line_requests",
           "requests_per_launch": round(reqs),
           "achieved_requests_per_s": round(ach, -6),
           "peak_requests_per_s": peak, "frac": round(ach / peak, 4),
           "floor_ms_at_peak": round(reqs / peak * 1e3, 3),
           "source": "profiles/" + prof["source"]}
    lat = prof.get("mean_latency_cycles")
    if lat:
        # requests x latency / (CUs x slots) cycles: the time the vector L1s'
        # outstanding-request capacity allows at this mean latency
        out["mean_latency_cycles"] = round(lat, 1)
        out["tcp_slot_floor_ms"] = round(
            reqs * lat / (NUM_CUS * TCP_SLOTS) / (L2_CLOCK_GHZ * 1e6), 3)
    return out


# ------------------------------------------------------ secondary measurements
def window_variants(S, torch, x, y, Mloc, Nglob, K, fmt_family):
    """roofline.variants: the banded members of the headline family (SURVEY
    8d: "report W = N and W = 2^20"), autotuned like the headline, 20
    event-timed launches each."""
    import numpy as np
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for tag, W in (("W=2^20", 1 << 20), ("W=2^17", 1 << 17)):
        try:
            dA = S.CsrDevice.generate(FAMILIES["random"], Mloc, Nglob, K, W, 0,
                                      MATRIX_SEED)
            dH = dA.to_hll(True)
            dA.release()
            best, _ = dH.autotune(x.data_ptr(), y.data_ptr())
            ms = dH.time(best, x.data_ptr(), y.data_ptr(), 3, 20, 0, 0,
                         stream=st)
            kname = "hll_" + S.HLL_KERNEL_LABELS[best]
            wl = workload_name("random", "hll", Mloc, Nglob, Mloc, K, W, W)
            tr, _ = measured_traffic(
                wl, kname, dH.panels_schedule()
                if best == S.HLL_KERNEL_PANELS else None)
            b = dH.kernel_bytes(best)
            out[tag] = {
                "kernel": kname,
                "layout": dH.panels_describe()
                if best == S.HLL_KERNEL_PANELS else None,
                "kernel_ms": round(float(np.mean(ms)), 5),
                "gflops": round(2.0 * dH.NZ / (float(np.mean(ms)) * 1e6), 1),
                "achieved": round(b / (float(np.mean(ms)) * 1e6), 1),
                "frac": round(b / (float(np.mean(ms)) * 1e6) / HBM_PEAK_GBPS, 4),
                "traffic": round(tr["bytes_per_launch"]) if tr else None,
                "profile": ("profiles/" + tr["source"]) if tr else None,
            }
            dH.release()
        except OSError as e:  # e.g. out of memory on a smaller card
            out[tag] = {"error": str(e)}
    return out


def extra_measurements(S, torch, mat, x, y, Mloc, Nglob, K):
    """compact secondary numbers, [kernel_ms, GFLOP/s, roofline fraction]
    per tag: the direct HLL kernels on the headline matrix (kernel 1 is the
    literal north-star form), the banded 10M matrix, and BASELINE config 2
    (1M banded CSR; 212 MB < Infinity Cache, so every launch follows a
    1 GiB read-only flush)."""
    import numpy as np
    st = torch.cuda.current_stream().cuda_stream
    dx, dy = x.data_ptr(), y.data_ptr()
    out = {}

    def row(tag, m, ms):
        ms = float(np.median(ms))
        out[tag] = [round(ms, 4), round(2.0 * m.NZ / (ms * 1e6), 1),
                    round(m.algorithmic_bytes / (ms * 1e6) / HBM_PEAK_GBPS, 4)]

    try:
        if hasattr(mat, "num_blocks") and mat.col_major:
            for k in (1, 2):
                row("W=N hll_%s" % S.HLL_KERNEL_NAMES[k], mat,
                    mat.time(k, dx, dy, 2, 8, 0, 0, stream=st))
        dA = S.CsrDevice.generate(FAMILIES["banded"], Mloc, Nglob, K, 0, 0,
                                  MATRIX_SEED)
        dH = dA.to_hll(True)
        row("banded10M hll_threads_col_major", dH,
            dH.time(1, dx, dy, 2, 10, 0, 0, stream=st))
        row("banded10M csr_stream", dA,
            dA.time(4, dx, dy, 2, 10, 0, 0, stream=st))
        dH.release()
        dA.release()
        dB = S.CsrDevice.generate(FAMILIES["banded"], 1_000_000, 1_000_000, 16,
                                  0, 0, MATRIX_SEED)
        for k in (1, 2, 4):
            row("config2 csr_%s flushed" % S.CSR_KERNEL_NAMES[k], dB,
                dB.time(k, dx, dy, 2, 20, FLUSH_BYTES, 0, stream=st))
        best, _ = dB.autotune(dx, dy, True)
        row("config2 autotuned csr_%s flushed" % S.CSR_KERNEL_LABELS[best],
            dB, dB.time(best, dx, dy, 2, 20, FLUSH_BYTES, 0, stream=st))
        # the reference's seam as it is called (host arrays in, host y out:
        # upload + ONE launch + download per call, cuda_csr.cu:210-234): the
        # PCIe-inclusive rate of the drop-in, never `value`
        hA = dB.download()
        xh = S.vec_synth(1_000_000, X_SEED)
        S.csr_spmv_hip(hA, xh, kernel=4)  # first call: allocations warm
        t0 = time.perf_counter()
        _, kms = S.csr_spmv_hip(hA, xh, kernel=4)
        wall = (time.perf_counter() - t0) * 1e3
        out["config2 one-shot seam, host in/out (PCIe incl.)"] = [
            round(wall, 2), round(2.0 * dB.NZ / (wall * 1e6), 1),
            "kernel %.4f ms of it" % kms]
        S.csr_free(hA)
        dB.release()
    except OSError as e:
        out["error"] = str(e)
    # ---- the other single-GPU BASELINE configs, driver-timed in this line:
    # config 4 through the real path (.mtx -> loader -> upload -> selector)
    # and one rank's shard of config 5 (10M rows x 80M columns)
    t0 = time.time()
    try:
        path, info = config4_file("", 160)
        A = S.io_load_csr_cached(path)
        M4, N4 = A.contents.M, A.contents.N
        dA = S.CsrDevice.upload(A)
        x4 = S.DevBuffer.from_numpy(S.vec_random(N4))  # the reference's x
        y4 = S.DevBuffer(M4 * 8)
        best, _ = dA.autotune(x4.ptr, y4.ptr)
        tune4 = round(time.time() - t0, 2)
        row("config4 %s %dx%d csr_%s" % (
            "nlpkkt160.mtx" if "generated" not in info["source"]
            else "nlpkkt160-shaped .mtx", M4, N4, S.CSR_KERNEL_LABELS[best]),
            dA, dA.time(best, x4.ptr, y4.ptr, 2, 10, 0, 0, stream=st))
        out["config4_setup_s"] = dict(info, load_upload_tune_s=tune4)
        for o in (dA, x4, y4):
            o._release_now()
        S.csr_free(A)
    except (OSError, subprocess.CalledProcessError) as e:
        out["config4 error"] = str(e)
    t1 = time.time()
    try:
        N5 = 8 * Mloc
        dA = S.CsrDevice.generate(FAMILIES["random"], Mloc, N5, K, 2 * N5,
                                  3 * Mloc, MATRIX_SEED)
        dH = dA.to_hll(True)
        dA.release()
        x5 = S.DevBuffer(N5 * 8)
        S.dev_fill_synth(x5.ptr, N5, X_SEED)
        best, _ = dH.autotune(x5.ptr, dy)
        row("config5 shard %dx%d hll_%s" % (Mloc, N5, S.HLL_KERNEL_LABELS[best]),
            dH, dH.time(best, x5.ptr, dy, 2, 10, 0, 0, stream=st))
        if best == S.HLL_KERNEL_PANELS:
            out["config5 shard layout"] = dH.panels_describe()
        dH.release()
        x5.free()
    except OSError as e:
        out["config5 error"] = str(e)
    out["configs_4_5_s"] = [round(t1 - t0, 1), round(time.time() - t1, 1)]
    # ---- the reference's irregular classes (scripts/download-matrices.py:
    # 7-38), autotuned CSR: power-law rows of mean 3 (webbase / amazon /
    # roadNet) and a dc1-like hub row of 131072 entries + hub column
    try:
        for tag, fam, M2, K2, W2 in (
                ("powerlaw 4Mx3 anywhere", "powerlaw", 4_000_000, 3, 8_000_000),
                ("hub 1Mx6 W=4096", "hub", 1_000_000, 6, 4096)):
            if M2 > Nglob:
                continue  # x / y of this run are too short (--rows-per-gpu)
            dA = S.CsrDevice.generate(FAMILIES[fam], M2, M2, K2, W2, 0,
                                      MATRIX_SEED)
            best, _ = dA.autotune(dx, dy)
            fl = FLUSH_BYTES if dA.algorithmic_bytes < (512 << 20) else 0
            row("%s csr_%s%s" % (tag, S.CSR_KERNEL_LABELS[best],
                                 " flushed" if fl else ""),
                dA, dA.time(best, dx, dy, 2, 10, fl, 0, stream=st))
            dA.release()
    except OSError as e:
        out["irregular error"] = str(e)
    return out


def config4_file(mtx, kkt_n):
    """-> (path, info) of BASELINE config 4's input: --mtx, else the real
    $SPMV_MTX_DIR/nlpkkt160.mtx when present, else the nlpkkt160-shaped file
    of tools/gen_kkt_mtx.c (written once into the temp directory)"""
    info = {}
    path = mtx
    real = os.path.join(os.environ.get("SPMV_MTX_DIR", ""), "nlpkkt160.mtx")
    if not path and os.environ.get("SPMV_MTX_DIR") and os.path.exists(real):
        path = real
    if not path:
        gen = os.path.join(ROOT, "spmv_scpa_amd", "bin", "gen_kkt_mtx")
        path = os.path.join(tempfile.gettempdir(), "spmv_kkt%d.mtx" % kkt_n)
        if not os.path.exists(path):
            t0 = time.time()
            subprocess.run([gen, str(kkt_n), path + ".part"],
                           check=True, capture_output=True)
            os.replace(path + ".part", path)
            info["mtx_write_s"] = round(time.time() - t0, 2)
        info["source"] = ("generated nlpkkt160-shaped KKT file "
                          "(tools/gen_kkt_mtx.c, %d^3 grid)" % kkt_n)
    else:
        info["source"] = path
    return path, info


# ------------------------------------------------------------------ config 4/2
def single_matrix_bench(args, S, torch, dev):
    """--config 4 (.mtx through the loader, CSR) and --config 2 (1M banded
    CSR, flushed): one GPU, one matrix, autotuned CSR kernel."""
    import numpy as np
    st = torch.cuda.current_stream().cuda_stream
    info = {}
    t_setup = time.time()
    if args.config == 4:
        path, info = config4_file(args.mtx, args.kkt_n)
        had_bin = os.path.exists(path + ".bin")
        t0 = time.time()
        A = S.io_load_csr_cached(path)
        info["load_s"] = round(time.time() - t0, 2)
        info["loaded_from"] = ".bin sidecar" if had_bin else \
            ".mtx text (sidecar written)"
        if not had_bin:
            S.csr_free(A)
            t0 = time.time()
            A = S.io_load_csr_cached(path)
            info["bin_load_s"] = round(time.time() - t0, 2)
        M, N, NZ = A.contents.M, A.contents.N, A.contents.NZ
        name = A.contents.name.decode()
        xh = S.vec_random(N)  # the reference's x for .mtx runs
        dA = S.CsrDevice.upload(A)
        x = torch.from_numpy(xh).to(dev)
        flush = 0
        workload = ("%s.mtx %dx%d, %d nnz after symmetric expansion, CSR "
                    "(BASELINE config 4: nlpkkt160; %s)"
                    % (name, M, N, NZ, info["source"]))
    else:
        M = N = 1_000_000
        A = None
        dA = S.CsrDevice.generate(FAMILIES["banded"], M, N, 16, 0, 0,
                                  MATRIX_SEED)
        NZ = dA.NZ
        x = torch.empty(N, dtype=torch.float64, device=dev)
        S.dev_fill_synth(x.data_ptr(), N, X_SEED, 0, st)
        flush = FLUSH_BYTES  # 212 MB working set < 256 MiB Infinity Cache
        workload = ("banded CSR 1000000x1000000, 16 nnz/row (BASELINE "
                    "config 2), 1 GiB read-only flush between launches")
    y = torch.zeros(M, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    if args.blocked_pin:  # the layout an earlier line printed (profiling)
        kernel, tuned = S.CSR_KERNEL_PANELS, None
        dA.build_panels_pinned(args.blocked_pin)
    elif args.kernel >= 0:
        kernel, tuned = args.kernel, None
        if kernel == S.CSR_KERNEL_PANELS:  # fixed: default chain layout (the
            dA.build_panels(0, "chain")    # SPMV_TILE_ROWS knob applies)
    else:
        t_tune = time.time()
        kernel, tuned = dA.autotune(x.data_ptr(), y.data_ptr())
        info["tune_s"] = round(time.time() - t_tune, 2)
    kname = "csr_" + S.CSR_KERNEL_LABELS[kernel]
    t_setup = time.time() - t_setup

    # result check: rows of y against the rows of the HOST matrix
    dA.launch(kernel, x.data_ptr(), y.data_ptr(), stream=st)
    torch.cuda.synchronize()
    rng = np.random.default_rng(1234)
    rows = np.concatenate([[0, M - 1], rng.integers(0, M, 256)])
    got = y[torch.as_tensor(rows, device=dev)].cpu().numpy()
    if A is not None:
        IRP, JA, AS = S.csr_arrays(A)
        xh_ = x.cpu().numpy()
        for g, r in zip(got, rows):
            c, v = JA[IRP[r]:IRP[r + 1]], AS[IRP[r]:IRP[r + 1]]
            t = v * xh_[c]
            if abs(g - t.sum()) > 1e-6 * max(abs(t.sum()), 1e-3 * np.abs(t).sum()):
                raise SystemExit("parity check failed on row %d" % r)
    else:
        check_rows(S, FAMILIES["banded"], N, 16, 0, got, rows)

    # warm-up + EXACTLY K timed steps (flushed between steps for config 2:
    # the flush is outside the per-step events, wall time is not the metric)
    for _ in range(args.warmup):
        dA.launch(kernel, x.data_ptr(), y.data_ptr(), stream=st)
    torch.cuda.synchronize()
    if flush:
        kern_ms = dA.time(kernel, x.data_ptr(), y.data_ptr(), 0, args.steps,
                          flush, args.waves, stream=st)
        ms_per_step = float(np.mean(kern_ms))
    else:
        ev = [(torch.cuda.Event(enable_timing=True),
               torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps)]
        t0 = time.perf_counter()
        for a, b in ev:
            a.record()
            dA.launch(kernel, x.data_ptr(), y.data_ptr(),
                      waves_per_block=args.waves, stream=st)
            b.record()
        torch.cuda.synchronize()
        ms_per_step = (time.perf_counter() - t0) * 1e3 / args.steps
        kern_ms = [a.elapsed_time(b) for a, b in ev]
    alg = dA.algorithmic_bytes
    out = {
        "metric": METRIC, "value": round(2.0 * NZ / (ms_per_step * 1e6), 2),
        "unit": "GFLOP/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic" if args.config == 2 or "generated" in
        info.get("source", "") else "file",
        "config": dict({"workload": workload, "kernel": kname,
                        "kernel_choice": "autotuned (spmv_csr_autotune)"
                        if tuned is not None else
                        "pinned layout (--blocked-pin)" if args.blocked_pin
                        else "fixed by --kernel",
                        "blocked_schedule": dA.panels_schedule()
                        if kernel == S.CSR_KERNEL_PANELS else None,
                        "blocked_layout": dA.panels_describe()
                        if kernel == S.CSR_KERNEL_PANELS else None,
                        "blocked_pin": dA.panels_pin()
                        if kernel == S.CSR_KERNEL_PANELS else None,
                        "kernel_source": kernel_source_ident(kname),
                        "rows": M, "nnz": NZ}, **info),
        "roofline": roofline_dict(alg, kern_ms, kname, NZ,
                                  *measured_traffic(
                                      workload, kname, dA.panels_schedule()
                                      if kernel == S.CSR_KERNEL_PANELS
                                      else None)),
        "host": {"host_gap_ms": round(ms_per_step - float(np.mean(kern_ms)), 5)
                 if not flush else None},
        "setup_s": round(t_setup, 2), "rows_checked": len(rows),
    }
    if not args.no_extras:
        ex = {}
        for k in (1, 2, 4):
            ms = float(np.median(dA.time(k, x.data_ptr(), y.data_ptr(), 2, 10,
                                         flush, args.waves, stream=st)))
            ex["csr_" + S.CSR_KERNEL_NAMES[k]] = [
                round(ms, 4), round(2.0 * NZ / (ms * 1e6), 1),
                round(alg / (ms * 1e6) / HBM_PEAK_GBPS, 4)]
        out["extras"] = ex
    if not args.no_cpu_baseline and args.config == 2:
        out["cpu_baseline"] = cpu_baseline(
            S, FAMILIES["banded"], M, N, 16, 0,
            args.cpu_csv_dir or os.path.join(ROOT, "gpurun_out", "cpu_baseline"),
            "banded1M")
    print(json.dumps(out))


# -------------------------------------------------------------------- launcher
def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def visible_gpus():
    """GPUs this job can use, counted WITHOUT loading a GPU runtime into this
    process (the parent only spawns; VERDICT r03 #9: torch.cuda.device_count()
    may initialise HIP): a short-lived CHILD asks torch (which honours
    ROCR_ / HIP_ / CUDA_VISIBLE_DEVICES and the container's device cgroup);
    should that fail, the KFD topology in sysfs is counted (nodes with SIMDs),
    clipped by the *_VISIBLE_DEVICES lists.  None: unknown -- the ranks then
    find out themselves."""
    try:
        r = subprocess.run(
            [sys.executable, "-c",
             "import torch; print(torch.cuda.device_count())"],
            capture_output=True, text=True, timeout=180)
        if r.returncode == 0:
            return int(r.stdout.strip().splitlines()[-1])
    except (OSError, ValueError, IndexError, subprocess.TimeoutExpired):
        pass
    return kfd_gpu_count()


def kfd_gpu_count(root="/sys/class/kfd/kfd/topology/nodes", env=None):
    """GPU nodes of the KFD topology (simd_count > 0), at most as many as a
    *_VISIBLE_DEVICES list names; None when sysfs has no KFD topology"""
    env = os.environ if env is None else env
    try:
        nodes = sorted(os.listdir(root))
    except OSError:
        return None
    n = 0
    for d in nodes:
        try:
            for line in open(os.path.join(root, d, "properties")):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
        except (OSError, ValueError):
            continue
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES",
                "CUDA_VISIBLE_DEVICES"):
        if env.get(var, "").strip():
            n = min(n, len([t for t in env[var].split(",") if t.strip()]))
    return n


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank
    processes of this script (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, rendezvous on 127.0.0.1), relay rank 0's JSON line, exit
    with the worst return code.  The parent never loads a GPU runtime (devices
    are counted by a child, visible_gpus), so nothing that initialised HIP is
    ever re-executed."""
    n = args.gpus
    if args.backend == "nccl":
        have = visible_gpus()
        if have is not None and have < n:
            sys.stderr.write("bench.py: --gpus %d but %d device(s) visible\n"
                             % (n, have))
            return 2
    env = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    # RCCL between processes needs dmabuf IPC on this pool's host driver
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(
            [sys.executable, os.path.abspath(__file__)] + list(argv), env=e,
            stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = b""
    worst, failed_at = 0, None
    live = set(range(n))
    import select
    while live:
        if 0 in live:  # keep rank 0's pipe drained
            rd, _, _ = select.select([procs[0].stdout], [], [], 0.2)
            if rd:
                chunk = os.read(procs[0].stdout.fileno(), 65536)
                out0 += chunk
        else:
            time.sleep(0.2)
        for r in list(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if r == 0:
                out0 += procs[0].stdout.read() or b""
            if rc != 0:
                worst = rc if worst == 0 or abs(rc) > abs(worst) else worst
                failed_at = failed_at or time.time()
        # a rank died: the others would wait in a collective forever
        if failed_at and time.time() - failed_at > 20:
            for r in live:
                procs[r].kill()  # exactly the children started above
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    return worst if worst >= 0 else 128 - worst


def rendezvous_only(args, rank, world):
    """--rendezvous-only: the launcher / process-group plumbing by itself"""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    gpu = args.backend == "nccl"
    if gpu:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(args.backend)
    t = torch.tensor([float(rank + 1), 1.0], device="cuda" if gpu else "cpu")
    dist.all_reduce(t)
    ok = float(t[0].item()) == world * (world + 1) / 2
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "GFLOP/s",
                          "n_gpus": world, "rendezvous_only": True,
                          "backend": args.backend, "ranks_joined": ok,
                          "nranks_joined": int(t[1].item())}))
    dist.destroy_process_group()
    return 0 if ok else 1


# ------------------------------------------------------------------------ main
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    omp_team = cap_openmp_env()
    if ("WORLD_SIZE" not in os.environ and args.gpus > 1
            and not args.native_mgpu):
        raise SystemExit(launch_ranks(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one rank per "
                         "GPU (python bench.py --gpus N starts them itself)"
                         % (args.gpus, world))
    if args.native_mgpu:
        return native_mgpu_bench(args, argv, omp_team)
    if args.rendezvous_only:
        raise SystemExit(rendezvous_only(args, rank, world))
    stat0 = cgroup_cpu_stat()

    import numpy as np
    import torch
    import torch.distributed as dist
    import spmv_scpa_amd as S
    from spmv_scpa_amd import dist as D

    if not torch.cuda.is_available() or S.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X: no GPU visible "
                         "(there is no CPU fallback)")
    if args.backend == "gloo":  # rehearsal: ranks may share a card
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    S.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.config != 3:
        if world > 1:
            raise SystemExit("--config %d is a single-GPU line" % args.config)
        return single_matrix_bench(args, S, torch, dev)
    use_dist = world > 1 or args.force_exchange
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    L = args.shards_per_gpu
    if args.strong:
        if 8 % world:
            raise SystemExit("--strong needs 1, 2, 4 or 8 GPUs")
        L = 8 // world
    Mshard, K = args.rows_per_gpu, args.nnz_row
    Mloc = Mshard * L  # rows of this rank
    Mglob = Mloc * world
    Nglob = Mglob
    W = args.window if args.window > 0 else 2 * Nglob  # >= 2N: anywhere
    kind = FAMILIES[args.family]
    row0 = rank * Mloc

    # ---- build the shard(s) in HBM (device-side generator + converter) ----
    t_setup = time.time()
    x = torch.empty(Nglob, dtype=torch.float64, device=dev)
    y = torch.zeros(Mglob, dtype=torch.float64, device=dev)
    S.dev_fill_synth(x.data_ptr(), Nglob, X_SEED, 0,
                     torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    tuned, t_tune = None, None

    def build_shards(count, rows, first_row=row0, ncols=Nglob, w=W):
        """`count` logical shards of `rows` rows starting at `first_row`"""
        out, nnz, stored = [], 0, 0
        for j in range(count):
            dA = S.CsrDevice.generate(kind, rows, ncols, K, w,
                                      first_row + j * rows, MATRIX_SEED)
            nnz += dA.NZ
            if args.format == "hll":
                col_major = True if args.kernel in (-1, 4) else \
                    S.HLL_KERNEL_COL_MAJOR[args.kernel]
                m = dA.to_hll(col_major)
                stored += m.slots
                dA.release()
            else:
                m = dA
                stored += dA.NZ
            out.append(m)
        return out, nnz, stored

    mats, nnz_local, slots = build_shards(L, Mshard)
    mat = mats[0]
    if args.format == "hll":
        labels, prefix = S.HLL_KERNEL_LABELS, "hll_"
    else:
        labels, prefix = S.CSR_KERNEL_LABELS, "csr_"
    if args.blocked_pin and use_dist:
        raise SystemExit("--blocked-pin pins ONE rank's layout for the "
                         "profiling passes: single GPU only")
    pinned = bool(args.blocked_pin)
    if pinned:
        kernel = (S.HLL_KERNEL_PANELS if args.format == "hll"
                  else S.CSR_KERNEL_PANELS)
        for m in mats:
            m.build_panels_pinned(args.blocked_pin)
    elif args.kernel >= 0:
        kernel = args.kernel
    else:
        # kernel chosen by measurement (spmv_*_autotune) on the first shard:
        # the coalesced kernels and, if they run far below the stream rate,
        # the 2-D blocked path.  Every rank must take the same decision:
        # rank 0's pick -- kernel id and, for the blocked path, its schedule
        # and tile height -- is broadcast (they decide how the exchange is
        # arranged below: every rank must issue the same collectives).
        t_tune = time.time()
        kernel, tuned = mat.autotune(x.data_ptr(), y.data_ptr() + 8 * row0)
        t_tune = time.time() - t_tune
        if use_dist:
            mine = D.Pick(kernel, mat.panels_schedule(),
                          mat.panels_tile_rows() or 0)
            pick = D.agree_on_pick(dist, mine, dev)
            kernel = pick.kernel
            if labels[kernel] == "tile_panels" and not pick.same_build(mine):
                mat.build_panels(0, pick.schedule, pick.tile_rows)
    blocked = labels[kernel] == "tile_panels"
    if blocked and mat.panels_info() is None:
        mat.build_panels(0)
    arrangement = None
    sweep = blocked and mat.panels_schedule() == "sweep"
    nsplit = 2 if args.force_exchange and world == 1 else 4
    if (blocked and not sweep and use_dist and L == 1
            and Mshard % (nsplit * D.HACK) == 0):
        # the blocked path runs whole matrices only: hold the rank's rows as
        # `nsplit` logical shards (4, like the row chunks of the direct
        # kernels) so that the all-gather of one shard runs under the kernel
        # of the next -- at 8 GPUs the exchange (560 MB in per GPU) is longer
        # than the kernel of a matrix with locality.
        model = mat
        for m in mats[1:]:
            m.release()
        L, Mshard = nsplit, Mshard // nsplit
        mats, nnz_local, slots = build_shards(L, Mshard)
        for m in mats:  # the tuned schedule and tile height
            m.build_panels_like(model)
        model.release()
        mat = mats[0]
        arrangement = "chain: %d logical shards, all-gather of shard c " \
                      "under the kernel of c+1" % L
    if blocked:
        for m in mats[1:]:  # the tuned shard's schedule and tile height
            if m.panels_info() is None:
                m.build_panels_like(mat)
    kname = prefix + labels[kernel]
    torch.cuda.synchronize()

    chunks = args.chunks if args.chunks > 0 else (4 if world > 1 else 1)
    if blocked or L > 1 or (args.format == "csr" and kernel == 4):
        chunks = 1  # the blocked path runs whole shards only; with logical
        #             shards the shard is the unit of overlap; the CSR stream
        #             kernel's row-block table covers the whole shard (a row
        #             sub-range would fall back to the sub-wave kernel)
    halo = 0
    if args.exchange == "halo":
        halo = args.halo_rows
        if halo <= 0:
            if args.family == "banded":
                halo = K
            elif args.window > 0 and args.family != "stencil":
                halo = (W + 1) // 2
            else:
                raise SystemExit("--exchange halo needs --halo-rows (or a "
                                 "column window)")
        halo = -(-halo // D.HACK) * D.HACK

    def make_sharded(ms, rows_total=Mloc, xx=x, yy=y):
        return D.ShardedSpmv(ms if len(ms) > 1 else ms[0], kernel, rank, world,
                             rows_total, xx, yy, waves_per_block=args.waves,
                             chunks=chunks,
                             force_exchange=args.force_exchange,
                             mode="halo" if halo else None, halo_rows=halo)

    sharded = make_sharded(mats)

    def time_steps(sh, n):
        """barrier-bracketed wall time of n steps, max over ranks, in ms/step"""
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            sh.step()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64,
                         device=dev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) * 1e3 / n

    if (sweep and use_dist and L == 1 and not halo
            and Mshard % (2 * D.HACK) == 0):
        # The sweep launch is persistent and wants its whole grid resident
        # (phase counters), so by default the exchange FOLLOWS the kernel.
        # Alternative: two logical shards, each swept by a grid that leaves
        # --reserve-cus compute units free, the all-gather of the first half
        # running beside the sweep of the second.  Whether RCCL's kernels and
        # the persistent grid share the chip well is a property of the node:
        # both arrangements are timed here (5 steps each, max over ranks) and
        # the faster one is kept -- the same "choose by measurement" rule as
        # the kernel selector, and every rank sees the same reduced times.
        try:
            alt, nnz_alt, slots_alt = build_shards(2, Mshard // 2)
            for m in alt:
                m.build_panels(0, "sweep", reserve_cus=args.reserve_cus)
            sh_alt = make_sharded(alt)
            for s_ in (sharded, sh_alt):
                s_.step()
            t_serial = time_steps(sharded, 5)
            t_split = time_steps(sh_alt, 5)
            arrangement = ("sweep: exchange after the kernel %.3f ms/step vs "
                           "2 logical shards on %d fewer CUs with overlapped "
                           "all-gather %.3f ms/step"
                           % (t_serial, args.reserve_cus, t_split))
            if t_split < t_serial:
                for m in mats:
                    m.release()
                mats, sharded, L, Mshard = alt, sh_alt, 2, Mshard // 2
                nnz_local, slots, mat = nnz_alt, slots_alt, alt[0]
                arrangement += " -> overlapped"
            else:
                for m in alt:
                    m.release()
                arrangement += " -> exchange after the kernel"
        except OSError as e:
            arrangement = "sweep: overlapped arrangement not built (%s)" % e
    pinfo = mat.panels_info() if blocked else None
    launches_per_step = (pinfo["steps"] if pinfo else 1) * L
    # per step and GPU; one launch per logical shard.  Priced
    # for the kernel that runs: the blocked copy of an HLL handle stores no
    # padding (spmv_hll_kernel_bytes); same number when the format pads nothing
    alg_bytes = sum(m.kernel_bytes(kernel) for m in mats)
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    # ---- result check: rows of y recomputed from the workload definition ----
    sharded.step()
    torch.cuda.synchronize()
    rng = np.random.default_rng(1234 + rank)
    rows = np.concatenate([[0, Mloc - 1], rng.integers(0, Mloc, 256)])
    if world > 1:  # and rows every OTHER rank computed: the exchange
        extra = []
        for r in range(world):
            if r == rank:
                continue
            if halo:  # only what lies within the halo of this rank's rows
                _, recv = sharded.halo_slices(r)
                if recv:
                    extra.append(np.array([recv[0], recv[1] - 1]) - row0)
            else:
                extra.append(np.array([0, Mloc // 2, Mloc - 1])
                             + (r - rank) * Mloc)
        rows = np.concatenate([rows] + extra)
    got = y[row0 + torch.as_tensor(rows, device=dev)].cpu().numpy()
    checked = check_rows(S, kind, Nglob, K, W, got, row0 + rows)
    stat1 = cgroup_cpu_stat()

    # ---- warm-up, then EXACTLY K timed steps ----
    for _ in range(args.warmup):
        sharded.step()

    def timed_steps():
        """K steps between barrier + synchronize on both sides; per-step
        events on the launch stream and host timestamps after each enqueue.
        -> (wall seconds, kernel ms per step, host seconds between enqueues)"""
        ev = [(torch.cuda.Event(enable_timing=True),
               torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps)]
        stamps = [0.0] * (args.steps + 1)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stamps[0] = t0
        for k in range(args.steps):
            sharded.step(events=ev[k])
            stamps[k + 1] = time.perf_counter()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        return (wall, [a.elapsed_time(b) for a, b in ev],
                [stamps[k + 1] - stamps[k] for k in range(args.steps)])

    stat2 = cgroup_cpu_stat()
    elapsed, kern_ms, enq = timed_steps()
    stat3 = cgroup_cpu_stat()
    attempts = [{"ms_per_step": round(elapsed * 1e3 / args.steps, 5),
                 "kernel_ms_avg": round(float(np.mean(kern_ms)), 5),
                 "max_enqueue_ms": round(max(enq) * 1e3, 4),
                 "throttled": stat_delta(stat2, stat3)}]
    # At N = 1 a step is one launch, so wall / step must equal the
    # event-timed kernel; a gap means the HOST stalled inside the timed
    # region (round 2: CFS throttling, 4.4 ms/step).  Then -- once, in the
    # same process -- K steps are timed again AS A DIAGNOSTIC
    # (host.retry_ms_per_step, top-level "host_stall_retry": true): `value`
    # always is the FIRST attempt, exactly K timed steps, never a best-of-two
    # (lines must stay comparable across rounds).
    gap = elapsed * 1e3 / args.steps - float(np.mean(kern_ms))
    retried = False
    if (world == 1 and not args.force_exchange
            and gap > 0.05 * float(np.mean(kern_ms))):
        e2, k2, q2 = timed_steps()
        stat4 = cgroup_cpu_stat()
        retried = True
        attempts.append({"ms_per_step": round(e2 * 1e3 / args.steps, 5),
                         "kernel_ms_avg": round(float(np.mean(k2)), 5),
                         "max_enqueue_ms": round(max(q2) * 1e3, 4),
                         "throttled": stat_delta(stat3, stat4),
                         "diagnostic_only": True})

    # the exchange by itself (kernel only / serial / overlapped)
    exch_ms = None
    if use_dist:
        for _ in range(2):
            sharded.exchange_only()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            sharded.exchange_only()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        exch_ms = (time.perf_counter() - t1) * 1e3 / 10

    # what joined, on which cards, and every rank's own kernel time
    rccl = per_rank = None
    if use_dist:
        rccl, per_rank = describe_job(S, torch, dist, dev, local_rank, world,
                                      args.backend, kern_ms)
    t = torch.tensor([elapsed, float(nnz_local)], dtype=torch.float64,
                     device=dev)
    if use_dist:
        tm = t[:1].clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        ts = t[1:].clone()
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)  # ragged / kkt: nnz differs
        elapsed, nnz_global = float(tm.item()), int(ts.item())
    else:
        nnz_global = nnz_local
    ms_per_step = elapsed * 1e3 / args.steps
    value = 2.0 * nnz_global / (ms_per_step * 1e6)

    # ---- N > 1: the fixed-problem reading of config 5 (80M x 80M, 8 logical
    # shards of 10M rows, 8/N per GPU), so that a scaling run can be read
    # against the ">= 6x y-throughput at 8 GPUs" target: rows/s of the SAME
    # problem at every N; the 1-GPU denominator is a committed measurement.
    strong = None
    if (world > 1 and not args.strong and not args.no_strong_leg
            and 8 % world == 0 and args.family == "random"
            and args.window <= 0):
        strong = strong_leg(args, S, D, torch, dist, dev, rank, world, kernel,
                            mat, blocked, build_shards, make_sharded,
                            time_steps, Mglob, Mshard * L)

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    workload = workload_name(args.family, args.format, Mloc, Nglob, Mglob, K,
                             args.window, W, L, Mshard)
    sched_now = mat.panels_schedule() if blocked else None
    traffic, why = (measured_traffic(workload, kname, sched_now) if world == 1
                    else (None, "single-GPU profiles only"))
    roof = roofline_dict(alg_bytes, kern_ms, kname, nnz_local, traffic, why)
    if per_rank:  # rank 0's events above; every rank's mean here
        roof["kernel_ms_per_rank"] = [round(v, 5) for v in per_rank]
        roof["kernel_ms_min_rank"] = round(min(per_rank), 5)
        roof["kernel_ms_max_rank"] = round(max(per_rank), 5)
    if world == 1 and sweep:  # the schedule for rows that reach beyond an L2
        roof["secondary"] = secondary_roofline(workload, kname,
                                               float(np.mean(kern_ms)),
                                               sched_now)
    out = {
        "metric": METRIC,
        "value": round(value, 2),
        "unit": "GFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True,
        "scaling": "strong" if args.strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "backend": ("gloo REHEARSAL (ranks share GPUs, host-staged "
                        "exchange: timings are not measurements)"
                        if args.backend == "gloo" else "nccl (RCCL)")
            if use_dist else None,
            "workload": workload,
            "kernel": kname,
            "kernel_choice": "pinned layout (--blocked-pin)" if pinned
            else "autotuned (spmv_%s_autotune)" % args.format
            if tuned is not None else "fixed by --kernel",
            # host seconds the selector took; its phase log when that is > 1 s
            "tune_s": round(t_tune, 2) if t_tune is not None else None,
            "tune_log": (mat.tune_log() or "").splitlines()
            if t_tune is not None and t_tune > 1.0 and L == 1
            and arrangement is None else None,
            "blocked_schedule": mat.panels_schedule() if blocked else None,
            "blocked_layout": mat.panels_describe() if blocked else None,
            # what --blocked-pin takes to run this layout again
            "blocked_pin": mat.panels_pin() if blocked else None,
            "kernel_source": kernel_source_ident(kname),
            "kernel_launches_per_step": launches_per_step,
            "rows_per_gpu": Mloc, "logical_shards_per_gpu": L,
            "nnz_per_row": K, "nnz_global": nnz_global,
            "stored_slots_per_gpu": slots,
            "partition": "contiguous row ranges, x replicated, in-place "
                         "all-gather(y) over RCCL" if world > 1 else "single GPU",
            "chunks": chunks, "exchange": sharded.mode,
            "exchange_arrangement": arrangement,
            "exchange_ms_alone": round(exch_ms, 5) if exch_ms else None,
            "rccl": rccl,
            "halo_rows": halo or None,
            "rows_per_s": round(Mglob / (ms_per_step * 1e-3), 1),
            "strong": strong,
        },
        "roofline": roof,
        "host": {
            "host_gap_ms": round(ms_per_step - float(np.mean(kern_ms)), 5),
            "max_enqueue_ms": round(max(enq) * 1e3, 4),
            "timing_attempts": attempts,
            "retry_ms_per_step": attempts[1]["ms_per_step"] if retried else None,
            "omp_team": omp_team,
            "cpu_quota": host_cpus()[1],
            # CFS periods / throttled periods of this cgroup: over the result
            # check, and over the whole run up to the end of the timed steps
            "cfs_check": stat_delta(stat0, stat1),
            "cfs_total": stat_delta(stat0, stat3),
        },
        "setup_s": round(t_setup, 2),
        "rows_checked": checked,
    }
    if retried:
        out["host_stall_retry"] = True
    # the >= 6x target is a FIXED-problem reading (80M x 80M on N GPUs vs 1):
    # top level, so a scaling run can be read without digging
    out["strong_speedup"] = strong_speedup_of(out, strong, world)
    single = world == 1 and L == 1 and not args.force_exchange
    if (single and not args.no_extras and args.family == "random"
            and args.window <= 0 and args.format == "hll"):
        roof["variants"] = window_variants(S, torch, x, y, Mloc, Nglob, K,
                                           args.family)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(
            S, kind, Mloc, Nglob, K, W,
            args.cpu_csv_dir or os.path.join(ROOT, "gpurun_out", "cpu_baseline"),
            "%s%dM" % (args.family, Mloc // 1_000_000))
    if single and not args.no_extras:
        out["extras"] = extra_measurements(S, torch, mat, x, y, Mloc, Nglob, K)
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


def describe_job(S, torch, dist, dev, local_rank, world, backend, kern_ms):
    """-> (config.rccl dict, [every rank's mean kernel ms]).  Collective: all
    ranks call it.  nranks_joined = an all-reduce of ones (what the
    communicator really spans), devices = PCI bus id per rank (two ranks on
    one card would show here), version = the RCCL torch drives."""
    import numpy as np
    ones = torch.ones(1, dtype=torch.float64, device=dev)
    dist.all_reduce(ones)
    mine = torch.tensor([float(np.mean(kern_ms))], dtype=torch.float64,
                        device=dev)
    allk = torch.zeros(world, dtype=torch.float64, device=dev)
    if backend == "gloo":  # rehearsal: no GPU all-gather in gloo
        host = torch.zeros(world, dtype=torch.float64)
        dist.all_gather_into_tensor(host, mine.cpu())
        allk = host
    else:
        dist.all_gather_into_tensor(allk, mine)
    try:
        bus = S.device_pci_bus_id(local_rank)
    except OSError:
        bus = "?"
    ids = [None] * world
    dist.all_gather_object(ids, bus)
    ver = None
    if backend == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001 - informational
            ver = None
    return ({"backend": "nccl (RCCL)" if backend == "nccl" else backend,
             "version": ver, "library_links": S.rccl_version(),
             "nranks_joined": int(round(float(ones.item()))),
             "devices": ids},
            [float(v) for v in allk.tolist()])


def strong_speedup_of(out, strong, world):
    """fixed 80M x 80M problem: ms on ONE GPU (committed measurement) / ms at
    this N.  From config.strong at 1 < N < 8, from this line itself when it IS
    the fixed problem (--strong, or N = 8 of the default workload); else None"""
    try:
        if strong and strong.get("speedup_vs_1gpu"):
            return strong["speedup_vs_1gpu"]
        one = None
        if strong and strong.get("one_gpu_ms_per_step") and world == 8:
            one = strong["one_gpu_ms_per_step"]
        elif out["scaling"] == "strong" and world > 1:
            one, _ = strong_one_gpu()
        if one:
            return round(one / out["ms_per_step"], 3)
    except (KeyError, TypeError):
        pass
    return None


def native_mgpu_bench(args, argv, omp_team):
    """--native-mgpu: the product library's own multi-GPU entry points
    (include/spmv_mgpu.h -> mgpu.hip: one process, ncclCommInitAll, a stream
    per device, every step = each device's shard kernel + ONE grouped in-place
    ncclAllGather of y), measured in the bench shape and printed in the same
    JSON as the torch.distributed path, so that whichever way a scaling run
    is taken, the library's own collective code is what was measured.
    Single process: `--gpus N` devices are driven from here, no ranks."""
    import numpy as np
    import spmv_scpa_amd as S
    n = args.gpus
    if S.device_count() < n:
        raise SystemExit("bench.py --native-mgpu --gpus %d: %d device(s) "
                         "visible (there is no CPU fallback)"
                         % (n, S.device_count()))
    if args.config != 3 or args.strong or args.shards_per_gpu != 1:
        raise SystemExit("--native-mgpu runs the default workload "
                         "(weak scaling, one shard per GPU)")
    kind = FAMILIES[args.family]
    Mloc, K = args.rows_per_gpu, args.nnz_row
    Mglob = Nglob = Mloc * n
    W = args.window if args.window > 0 else 2 * Nglob
    t_setup = time.time()
    g = S.MultiGpu(n)
    g.generate(kind, Mloc, K, W, MATRIX_SEED, as_hll=args.format == "hll")
    g.fill_x(X_SEED)
    # exchange: ONE grouped in-place all-gather after the shard kernels.  The
    # chunked, overlapped "staged" mode is opt-in (--chunks k): it has only
    # ever run as a 1-rank collective (spmv_mgpu.h), and the first real N > 1
    # run should not go down the most complex branch by default
    chunks = args.chunks if args.chunks > 0 else 1
    g.set_exchange(chunks, args.force_exchange)
    labels, prefix = ((S.HLL_KERNEL_LABELS, "hll_") if args.format == "hll"
                      else (S.CSR_KERNEL_LABELS, "csr_"))
    t_tune = None
    if args.kernel >= 0:
        kernel = args.kernel
    else:
        t_tune = time.time()
        kernel = g.autotune()
        t_tune = time.time() - t_tune
    kname = prefix + labels[kernel]
    t_setup = time.time() - t_setup

    # result check on what EVERY device holds after the exchange
    g.spmv(kernel, 0, 1)
    rng = np.random.default_rng(1234)
    rows = np.unique(np.concatenate(
        [[0, Mglob - 1], rng.integers(0, Mglob, 256)]
        + [np.array([0, Mloc // 2, Mloc - 1]) + r * Mloc for r in range(n)]))
    checked = 0
    for r in range(n):
        y = g.get_y(r)
        checked += check_rows(S, kind, Nglob, K, W, y[rows], rows)
        del y

    wall_ms, kms = g.run(kernel, args.warmup, args.steps)
    exch = g.exchange_only(10) if n > 1 else None
    ngp, _, nnz_global, alg_bytes = g.info()
    stored, _, layout = g.shard_info(0)
    if args.format == "hll" and kernel == S.HLL_KERNEL_PANELS:
        # the blocked copy stores the true entries, not the padded slots
        alg_bytes -= 12 * (stored - nnz_global // n)
    ms_per_step = wall_ms / args.steps
    workload = workload_name(args.family, args.format, Mloc, Nglob, Mglob, K,
                             args.window, W)
    traffic, why = (measured_traffic(workload, kname) if n == 1
                    else (None, "single-GPU profiles only"))
    roof = roofline_dict(alg_bytes, [float(np.mean(kms))], kname,
                         nnz_global // n, traffic, why)
    roof["kernel_ms_per_rank"] = [round(float(v), 5) for v in kms]
    roof["kernel_ms_min_rank"] = round(float(np.min(kms)), 5)
    roof["kernel_ms_max_rank"] = round(float(np.max(kms)), 5)
    out = {
        "metric": METRIC,
        "value": round(2.0 * nnz_global / (ms_per_step * 1e6), 2),
        "unit": "GFLOP/s", "n_gpus": n, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "backend": "native: one process, spmv_mgpu_* (mgpu.hip: "
                       "ncclCommInitAll, grouped in-place ncclAllGather)",
            "workload": workload, "kernel": kname,
            "kernel_choice": "autotuned (spmv_mgpu_autotune: device 0's pick "
                             "for all)" if t_tune is not None
            else "fixed by --kernel",
            "tune_s": round(t_tune, 2) if t_tune is not None else None,
            "blocked_layout": layout or None,
            "kernel_source": kernel_source_ident(kname),
            "kernel_launches_per_step": 1,
            "rows_per_gpu": Mloc, "logical_shards_per_gpu": 1,
            "nnz_per_row": K, "nnz_global": nnz_global,
            "stored_slots_per_gpu": stored,
            "partition": "contiguous row ranges, x replicated, in-place "
                         "all-gather(y) over RCCL" if n > 1 else "single GPU",
            "chunks": chunks,
            "exchange": ("staged: %d chunks, all-gather of chunk c under the "
                         "kernel of c+1" % chunks)
            if chunks > 1 and labels[kernel] not in ("tile_panels", "stream")
            and Mloc % (chunks * 32) == 0 and (n > 1 or args.force_exchange)
            else "allgather (after the kernels; one group)",
            "exchange_ms_alone": round(exch, 5) if exch else None,
            "rccl": {"backend": "RCCL as linked by libspmv_scpa_amd.so",
                     "version": S.rccl_version(),
                     "nranks_joined": g.comm_ranks(),
                     "devices": g.bus_ids()},
            "rows_per_s": round(Mglob / (ms_per_step * 1e-3), 1),
            "strong": None,
        },
        "roofline": roof,
        "host": {"host_gap_ms": round(ms_per_step - float(np.max(kms)), 5)
                 if n == 1 else None, "omp_team": omp_team,
                 "cpu_quota": host_cpus()[1]},
        "setup_s": round(t_setup, 2), "rows_checked": checked,
        "strong_speedup": None,
    }
    g.destroy()
    print(json.dumps(out))


def strong_one_gpu():
    """(ms per step, source) of the fixed 80M x 80M problem on ONE MI355X:
    the newest committed `bench.py --strong --gpus 1` line under profiles/
    (profiles/r*_strong_1gpu.json) whose blocked-kernel source is the tree's;
    a stale or missing file gives (None, why) and no speed-up is printed."""
    import glob
    _, blob = kernel_source_blob("hll_tile_panels")
    why = "no profiles/*_strong_1gpu.json committed"
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles",
                                            "*_strong_1gpu.json")),
                     reverse=True):
        try:
            j = json.load(open(fn))
            ks = j["config"].get("kernel_source") or {}
            if j["scaling"] != "strong" or j["n_gpus"] != 1:
                continue
            if not same_build(ks, "hll_tile_panels"):
                why = ("profiles/%s was measured with another build of "
                       "panels.hip / hip_common.h" % os.path.basename(fn))
                continue
            return float(j["ms_per_step"]), "profiles/" + os.path.basename(fn)
        except (ValueError, KeyError, OSError):
            continue
    return None, why


def strong_leg(args, S, D, torch, dist, dev, rank, world, kernel, model,
               blocked, build_shards, make_sharded, time_steps, Mglob_weak,
               _unused):
    """The fixed 80M x 80M problem at this N: 8/N logical shards of 10M rows
    per GPU with global columns, built with rank 0's pick.  At N = 8 this IS
    the weak-scaling workload (one shard per GPU), so nothing is rebuilt.
    Returns a small dict; any failure is reported, never fatal."""
    rows, total = args.rows_per_gpu, 8 * args.rows_per_gpu
    try:
        # the committed denominator is the FULL-size problem's
        one_ms, one_src = (strong_one_gpu() if rows == ROWS_PER_GPU else
                           (None, "not the 10M-rows-per-shard problem"))
        if world == 8 and Mglob_weak == total:
            return {"problem": "80M x 80M, 8 shards of 10M rows: identical to "
                               "this line's workload at N = 8",
                    "one_gpu_ms_per_step": one_ms,
                    "one_gpu_source": one_src,
                    "note": "speedup vs 1 GPU = one_gpu_ms_per_step / "
                            "ms_per_step of this line"}
        per = 8 // world
        xs = torch.empty(total, dtype=torch.float64, device=dev)
        ys = torch.zeros(total, dtype=torch.float64, device=dev)
        S.dev_fill_synth(xs.data_ptr(), total, X_SEED, 0,
                         torch.cuda.current_stream().cuda_stream)
        ms_, _, _ = build_shards(per, rows, rank * per * rows, total, 2 * total)
        if blocked:
            for m in ms_:
                m.build_panels_like(model)
        sh = make_sharded(ms_, per * rows, xs, ys)
        sh.step()
        ms = time_steps(sh, 5)
        for m in ms_:
            m.release()
        del xs, ys
        return {"problem": "80M x 80M fixed, %d logical shards of 10M rows "
                           "per GPU" % per,
                "ms_per_step": round(ms, 4),
                "rows_per_s": round(total / (ms * 1e-3), 1),
                "one_gpu_ms_per_step": one_ms, "one_gpu_source": one_src,
                "speedup_vs_1gpu": round(one_ms / ms, 3) if one_ms else None}
    except Exception as e:  # noqa: BLE001 - secondary figure
        return {"error": repr(e)}


if __name__ == "__main__":
    main()

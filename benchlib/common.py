"""Shared pieces of the bench modes: the workload constants, the CLI, the
result check against the product's HOST generator (never the oracle), the
roofline records and the lookups of committed rocprofv3 profiles.

bench.py (repo root) is the CLI; benchlib.single / dist / native are its modes."""
import argparse
import json
import os
import subprocess
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
                         "AS THE MAIN ARRANGEMENT (0 = the plain one: whole "
                         "shard, then one exchange; the chunked form is then "
                         "an optional leg reported as value_best)")
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
    ap.add_argument("--partition", default="even", choices=["even", "nnz"],
                    help="N>1: row ranges of equal row counts (default) or of "
                         "near-equal ENTRY counts (32-aligned cuts; the "
                         "multi-GPU form of the reference's "
                         "partition_csr_rows): ragged y fragments")
    ap.add_argument("--ragged-exchange", default="p2p",
                    choices=["p2p", "bcast", "padded"],
                    help="--partition nnz: grouped send/recv of exactly each "
                         "fragment (default), one broadcast per rank, or one "
                         "all-gather padded to the longest fragment + "
                         "compaction")
    ap.add_argument("--no-partition-leg", action="store_true",
                    help="N>1: skip config.partition_kkt (the nlpkkt160-"
                         "shaped matrix over the ranks, even vs nnz-balanced)")
    ap.add_argument("--no-family-leg", action="store_true",
                    help="N>1: skip config.family_variants (the banded and "
                         "the W = 2^20 members of the family through the same "
                         "plain path at this N)")
    ap.add_argument("--no-native-leg", action="store_true",
                    help="N>1: skip `native` (the library's own multi-GPU "
                         "path in a child process after the ranks are done)")
    ap.add_argument("--no-arrangement-choice", action="store_true",
                    help="N>1: skip the optional leg that builds and times "
                         "the overlapped arrangement (logical shards / row "
                         "chunks) beside the plain one (config.arrangements, "
                         "value_best); the line's `value` is the plain "
                         "arrangement's either way")
    ap.add_argument("--native-rehearsal", action="store_true",
                    help="--native-mgpu on a REHEARSAL handle: --gpus N "
                         "logical devices on the visible card(s), copies "
                         "instead of RCCL collectives (1-GPU test boxes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    return ap.parse_args(argv)


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
        # (bench lines print "same" here when it equals config.blocked_layout)
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


def last_json_line(path):
    """the record of a saved bench run: its LAST line that is a JSON object
    (a run prints its line provisionally after the main measurement and
    again, complete, at the end; older records are one pretty-printed or
    one-line object)"""
    text = open(path).read()
    for line in reversed(text.splitlines()):
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                break
    return json.loads(text)


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
            j = last_json_line(fn)
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


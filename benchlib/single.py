"""Single-GPU modes and the secondary measurements of the default line:
`--config 2` / `--config 4`, roofline.variants, extras."""
import json
import os
import subprocess
import time

from .common import *  # noqa: F401,F403
from .cpu import cpu_baseline


# ------------------------------------------------------ secondary measurements
def window_variants(S, torch, x, y, Mloc, Nglob, K, fmt_family):
    """roofline.variants: the other members of the headline family that
    SURVEY 8d asks to report beside W = N -- the banded windows W = 2^20 and
    W = 2^17, and the padded variant (row length uniform in [24, 40], mean 32:
    the HLL format stores S > nnz slots) -- autotuned like the headline, 20
    event-timed launches each.  A direct HLL kernel is priced on the STORED
    slots (it streams the padding), the blocked copy on the true entries (it
    stores none): spmv_hll_kernel_bytes."""
    import numpy as np
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for tag, fam, W in (("W=2^20", "random", 1 << 20),
                        ("W=2^17", "random", 1 << 17),
                        ("ragged[24,40] W=N", "ragged", 2 * Nglob),
                        ("ragged[24,40] W=2^17", "ragged", 1 << 17)):
        try:
            dA = S.CsrDevice.generate(FAMILIES[fam], Mloc, Nglob, K, W, 0,
                                      MATRIX_SEED)
            dH = dA.to_hll(True)
            dA.release()
            best, _ = dH.autotune(x.data_ptr(), y.data_ptr())
            ms = dH.time(best, x.data_ptr(), y.data_ptr(), 3, 20, 0, 0,
                         stream=st)
            kname = "hll_" + S.HLL_KERNEL_LABELS[best]
            wl = workload_name(fam, "hll", Mloc, Nglob, Mloc, K,
                               0 if W >= 2 * Nglob else W, W)
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
            }
            if tr:  # committed rocprofv3 passes of this variant
                out[tag]["traffic"] = round(tr["bytes_per_launch"])
                out[tag]["profile"] = "profiles/" + tr["source"]
            if fam == "ragged":  # what the format pads, and what was priced
                out[tag]["stored_slots_over_nnz"] = round(dH.slots / dH.NZ, 4)
                out[tag]["priced_on"] = (
                    "true entries (the blocked copy stores no padding)"
                    if best == S.HLL_KERNEL_PANELS else
                    "stored slots (a direct kernel streams the padding)")
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
        # ... and with the opt-in "keep the last upload" of the seam
        # (spmv_seam_cache(2), hip_csr.h): the matrix and x stay on the
        # device, a call is memset(y) + launch + 8 MB of y back
        S.seam_cache(2)
        try:
            import ctypes as C
            S.csr_spmv_hip(hA, xh, kernel=4)  # uploads, keeps
            # ONE caller buffer for y, the C entry point itself: a fresh 8 MB
            # numpy array per call is page faults, not the seam's time
            yb = np.full(1_000_000, -1.0)
            xp = xh.ctypes.data_as(C.POINTER(C.c_double))
            yp = yb.ctypes.data_as(C.POINTER(C.c_double))
            t0 = time.perf_counter()
            for _ in range(5):
                kms = S._lib.csr_spmv_hip_stream(hA, xp, yp, None)
            wall = (time.perf_counter() - t0) * 1e3 / 5
        finally:
            S.seam_cache(0)
        out["config2 one-shot seam, seam cache on"] = [
            round(wall, 3), round(2.0 * dB.NZ / (wall * 1e6), 1),
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


"""`--native-mgpu`: the product library's own multi-GPU entry points
(include/spmv_mgpu.h) measured in the bench shape, one process."""
import json
import os
import subprocess
import sys
import time

from .common import *  # noqa: F401,F403


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
    if S.device_count() < (1 if args.native_rehearsal else n):
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
    # --native-rehearsal: n LOGICAL devices on the visible card(s), copies
    # instead of collectives (spmv_mgpu.h): the control flow of an N > 1 run
    # on a 1-GPU box; its timings mean nothing
    g = S.MultiGpu(n, rehearsal=args.native_rehearsal)
    g.set_ragged_exchange(args.ragged_exchange)
    g.generate(kind, Mloc, K, W, MATRIX_SEED, as_hll=args.format == "hll",
               partition=args.partition)
    starts, nnz_per_rank, ragged = g.partition()
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
        if labels[kernel] == "tile_panels":  # fixed: default layout
            g.build_panels()
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
        + [np.array([starts[r], (starts[r] + starts[r + 1]) // 2,
                     starts[r + 1] - 1]) for r in range(n)
           if starts[r + 1] > starts[r]]))
    checked = 0
    for r in range(n):
        y = g.get_y(r)
        checked += check_rows(S, kind, Nglob, K, W, y[rows], rows)
        del y

    wall_ms, kms = g.run(kernel, args.warmup, args.steps)
    ngp, _, nnz_global, _ = g.info()
    stored, alg_bytes, layout = g.shard_info(0)
    if args.format == "hll" and kernel == S.HLL_KERNEL_PANELS:
        # the blocked copy stores the true entries, not the padded slots
        alg_bytes -= 12 * (stored - nnz_per_rank[0])
    ms_per_step = wall_ms / args.steps
    workload = workload_name(args.family, args.format, Mloc, Nglob, Mglob, K,
                             args.window, W)
    traffic, why = (measured_traffic(workload, kname) if n == 1
                    else (None, "single-GPU profiles only"))
    # device 0's shard against device 0's kernel time
    roof = roofline_dict(alg_bytes, [float(kms[0])], kname, nnz_per_rank[0],
                         traffic, why)
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
            "backend": "native REHEARSAL: one process, spmv_mgpu_* on %d "
                       "logical devices of the visible card(s), copies "
                       "instead of collectives: timings are not measurements"
                       % n if args.native_rehearsal else
                       "native: one process, spmv_mgpu_* (mgpu.hip: "
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
            "exchange_arrangement": "plain: every device's kernel, then ONE "
                                    "exchange of y" if n > 1 or
            args.force_exchange else None,
            "arrangements": None,
            "nnz_per_row": K, "nnz_global": nnz_global,
            "stored_slots_per_gpu": stored,
            "partition": ("nnz-balanced contiguous row ranges (32-aligned; "
                          "reference csr.c:218-276), ragged fragments by %s"
                          % args.ragged_exchange if ragged else
                          "contiguous row ranges of equal row counts, in-place "
                          "all-gather(y)") + ", x replicated, RCCL"
            if n > 1 else "single GPU",
            "row_starts": starts if ragged else None,
            "nnz_per_rank": nnz_per_rank if n > 1 else None,
            "chunks": chunks,
            "exchange":
            ("staged: %d chunks, all-gather of chunk c under the "
             "kernel of c+1" % chunks)
            if chunks > 1 and labels[kernel] not in ("tile_panels", "stream")
            and Mloc % (chunks * 32) == 0 and (n > 1 or args.force_exchange)
            and not ragged
            else "%s (after the kernels; one group)"
            % (args.ragged_exchange if ragged else "allgather"),
            "exchange_ms_alone": None,
            "exchange_alternatives_ms": None,
            "rccl": {"backend": "RCCL as linked by libspmv_scpa_amd.so",
                     "version": S.rccl_version(),
                     "nranks_joined": g.comm_ranks(),
                     "devices": g.bus_ids()},
            "rows_per_s": round(Mglob / (ms_per_step * 1e-3), 1),
            "strong": None,
            "rocm": S.rocm_runtime_report(),
        },
        "roofline": roof,
        "host": {"host_gap_ms": round(ms_per_step - float(np.max(kms)), 5)
                 if n == 1 else None, "omp_team": omp_team,
                 "cpu_quota": host_cpus()[1]},
        "setup_s": round(t_setup, 2), "rows_checked": checked,
        "strong_speedup": None,
    }
    # ---- the line goes out NOW (plain arrangement: every device's kernel,
    # then ONE grouped in-place all-gather -- what BASELINE names); the
    # alternative arrangements below cannot cost it any more
    multi = n > 1 or args.force_exchange
    arrange = (multi and not ragged and chunks == 1
               and Mloc % (4 * 32) == 0 and not args.no_arrangement_choice)
    pending = (["exchange_alone"] if multi else []) + \
        (["arrangement"] if arrange else [])
    failed, secs = [], {}
    if pending:
        print(json.dumps(dict(out, provisional=True, legs_pending=pending)),
              flush=True)
        if os.environ.get("SPMV_BENCH_NATIVE_HANG_AFTER_LINE"):
            time.sleep(3600)  # tests: a child that never finishes its legs
    if multi:
        t0 = time.time()
        try:
            out["config"]["exchange_ms_alone"] = round(g.exchange_only(10), 5)
            if not args.native_rehearsal:
                # the same fragments by the other ways the library can move
                # them (spmv_mgpu.h): the copy engines, and -- ragged
                # fragments -- the three RCCL forms
                alt = out["config"]["exchange_alternatives_ms"] = {}
                g.set_exchange_engine("copy")
                alt["copy"] = round(g.exchange_only(5), 5)
                g.set_exchange_engine("rccl")
                for kind_x in ("p2p", "bcast", "padded") if ragged else ():
                    g.set_ragged_exchange(kind_x)
                    alt[kind_x] = round(g.exchange_only(5), 5)
                if ragged:
                    g.set_ragged_exchange(args.ragged_exchange)
                else:
                    alt["allgather"] = round(g.exchange_only(5), 5)
        except Exception as e:  # noqa: BLE001 - an optional leg
            failed.append("exchange_alone: %r" % (e,))
            try:  # whatever failed, the handle goes back to its main engine
                if not args.native_rehearsal:
                    g.set_exchange_engine("rccl")
            except OSError:
                pass
        secs["exchange_alone"] = round(time.time() - t0, 1)
    if arrange and not failed:
        t0 = time.time()
        try:
            rec = native_arrangements(args, S, g, kernel, kind, Mloc, K, W,
                                      labels, layout)
            out["config"]["arrangements"] = rec
            best = rec.get("best_ms_per_step") or ms_per_step
            out["ms_per_step_best"] = round(best, 5)
            out["value_best"] = round(2.0 * nnz_global / (best * 1e6), 2)
            out["best_arrangement"] = rec["winner"]
        except Exception as e:  # noqa: BLE001 - an optional leg
            failed.append("arrangement: %r" % (e,))
        secs["arrangement"] = round(time.time() - t0, 1)
    if pending:
        out["legs_failed"], out["legs_s"] = failed, secs
    g.destroy()
    print(json.dumps(out), flush=True)


def native_arrangements(args, S, g, kernel, kind, Mloc, K, W, labels, layout):
    """Optional leg of --native-mgpu: the overlapped arrangements beside the
    plain one.  (b) the rows of a device as LOGICAL SHARDS (4; a sweep pick:
    2, on a grid that leaves --reserve-cus CUs to RCCL's kernels), shard c
    all-gathered on a second stream while shard c+1 computes.  (c) the same
    shards with the COPY ENGINE: every device pushes shard c into its peers' y
    with peer copies (SDMA over xGMI) -- no kernel competes with the SpMV for
    CUs, so no CUs are reserved.  Each is built beside the plain handle, timed
    over 10 steps against 10 steps of the plain one, and -- when it wins by
    3 % -- over exactly K steps (`best_ms_per_step`).  Every extra handle is
    destroyed before the next is built, whatever happens."""
    n = args.gpus
    sweep = (layout or "").startswith("sweep")
    L2 = 2 if sweep else 4
    g.spmv(kernel, 1, 1)
    t_plain = g.run(kernel, 1, 10)[0] / 10
    rec = {"plain_ms_per_step": round(t_plain, 5), "margin": 0.97,
           "alternatives": [], "winner": "plain"}
    cands = [] if args.native_rehearsal else [
        ("%d logical shards%s, RCCL all-gather of shard c under the kernel of "
         "c+1" % (L2, " on %d fewer CUs" % args.reserve_cus if sweep else ""),
         L2, args.reserve_cus if sweep else 0, "rccl")]
    cands.append(("%d logical shards, shard c pushed by the copy engines "
                  "under the kernel of c+1" % L2, L2, 0, "copy"))
    best = t_plain
    for label, L2_, res, eng in cands:
        g2 = S.MultiGpu(n, rehearsal=args.native_rehearsal)
        try:
            g2.set_logical_shards(L2_, res)
            if not args.native_rehearsal:
                g2.set_exchange_engine(eng)
            g2.generate(kind, Mloc, K, W, MATRIX_SEED,
                        as_hll=args.format == "hll")
            g2.fill_x(X_SEED)
            g2.set_exchange(1, args.force_exchange)
            if args.kernel >= 0:
                k2 = args.kernel
                if labels[k2] == "tile_panels":
                    g2.build_panels()
            else:
                k2 = g2.autotune()
            g2.spmv(k2, 1, 1)
            t2 = g2.run(k2, 1, 10)[0] / 10
            one = {"arrangement": label, "ms_per_step": round(t2, 5)}
            if t2 < 0.97 * best:
                tk = g2.run(k2, args.warmup, args.steps)[0] / args.steps
                one["ms_per_step_K"] = round(tk, 5)
                if tk < best:
                    best = tk
                    rec.update(winner=label, best_ms_per_step=round(tk, 5))
            rec["alternatives"].append(one)
        finally:
            g2.destroy()
    return rec


def native_leg(args, n, timeout_s=240):
    """`native` of an N > 1 line: a FRESH child process runs `bench.py
    --native-mgpu --gpus N` on the same workload (it needs the devices' HBM:
    the caller has released its own) and the essentials of its line come
    back.  Raises on failure; the caller records that in legs_failed."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--native-mgpu",
           "--gpus", str(n), "--steps", str(args.steps), "--warmup",
           str(args.warmup), "--rows-per-gpu", str(args.rows_per_gpu),
           "--nnz-row", str(args.nnz_row), "--window", str(args.window),
           "--family", args.family, "--format", args.format,
           "--kernel", str(args.kernel), "--partition", args.partition,
           "--ragged-exchange", args.ragged_exchange,
           "--no-cpu-baseline", "--no-extras"]
    if args.backend == "gloo":  # a rehearsal of the ranks: rehearse this too
        cmd.append("--native-rehearsal")
    if args.chunks > 0:
        cmd += ["--chunks", str(args.chunks)]
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE",
                        "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK",
                        "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    if timeout_s < 120:  # a short budget: the plain arrangement only
        cmd.append("--no-arrangement-choice")
    t0 = time.time()
    timed_out = False
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, env=env,
                           timeout=timeout_s)
        stdout = r.stdout
        if r.returncode != 0 and "{" not in stdout:
            raise RuntimeError("native child rc %d: %s" % (r.returncode,
                                                           r.stderr[-400:]))
    except subprocess.TimeoutExpired as e:
        # the child prints its line before its own optional legs: keep it
        stdout = e.stdout or ""
        if isinstance(stdout, bytes):
            stdout = stdout.decode(errors="replace")
        timed_out = True
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    if not lines:
        raise RuntimeError("native child printed no line within %.0f s"
                           % timeout_s)
    j = json.loads(lines[-1])
    c, roof = j["config"], j["roofline"]
    return {"backend": c["backend"], "value": j["value"],
            "exchange_arrangement": c.get("exchange_arrangement"),
            "ms_per_step": j["ms_per_step"], "kernel": c["kernel"],
            "blocked_layout": c.get("blocked_layout"),
            "kernel_ms_per_rank": roof.get("kernel_ms_per_rank"),
            "exchange": c.get("exchange"),
            "exchange_ms_alone": c.get("exchange_ms_alone"),
            "exchange_alternatives_ms": c.get("exchange_alternatives_ms"),
            "nnz_per_rank": c.get("nnz_per_rank"),
            "rccl": c.get("rccl"), "rows_checked": j.get("rows_checked"),
            "arrangements": c.get("arrangements"),
            "value_best": j.get("value_best"),
            "best_arrangement": j.get("best_arrangement"),
            "provisional": bool(j.get("provisional")) or None,
            "timed_out_after_s": timeout_s if timed_out else None,
            "legs_failed": j.get("legs_failed"),
            "wall_s": round(time.time() - t0, 1)}

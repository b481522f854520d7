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
    # ---- the exchange arrangement, chosen by measurement like the torch
    # path's.  (a) every device's kernel, then ONE RCCL all-gather (what
    # BASELINE names; always the first candidate).  (b) the rows of a device
    # as LOGICAL SHARDS (4; a sweep pick: 2, on a grid that leaves
    # --reserve-cus CUs to RCCL's kernels), shard c all-gathered on a second
    # stream while shard c+1 computes.  (c) the same shards with the COPY
    # ENGINE: every device pushes shard c into its peers' y with peer copies
    # (SDMA over xGMI) -- no kernel competes with the SpMV for CUs, so no CUs
    # are reserved.  Each is built and timed (5 steps); the fastest runs the K
    # timed steps.
    arrangement = None
    L_used, engine_used = 1, "copy" if args.native_rehearsal else "rccl"
    if ((n > 1 or args.force_exchange) and not ragged and chunks == 1
            and Mloc % (4 * 32) == 0 and not args.no_arrangement_choice):
        try:
            _, _, lay = g.shard_info(0)
            sweep = lay.startswith("sweep")
            L2 = 2 if sweep else 4
            g.spmv(kernel, 1, 1)
            best_t = g.run(kernel, 1, 5)[0] / 5
            notes = ["exchange after the kernels %.3f ms/step" % best_t]
            cands = [] if args.native_rehearsal else [
                ("%d logical shards%s, RCCL all-gather of shard c under the "
                 "kernel of c+1" % (L2, " on %d fewer CUs" % args.reserve_cus
                                    if sweep else ""),
                 L2, args.reserve_cus if sweep else 0, "rccl")]
            cands.append(("%d logical shards, shard c pushed by the copy "
                          "engines under the kernel of c+1" % L2, L2, 0,
                          "copy"))
            chosen = "exchange after the kernels"
            for label, L2_, res, eng in cands:
                g2 = S.MultiGpu(n, rehearsal=args.native_rehearsal)
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
                t2 = g2.run(k2, 1, 5)[0] / 5
                notes.append("%s %.3f ms/step" % (label, t2))
                if t2 < best_t:
                    g.destroy()
                    g, kernel, best_t = g2, k2, t2
                    L_used, engine_used, chosen = L2_, eng, label
                    starts, nnz_per_rank, ragged = g.partition()
                else:
                    g2.destroy()
            arrangement = " vs ".join(notes) + " -> " + chosen
        except OSError as e:
            arrangement = "alternative arrangements not built (%s)" % e
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
    exch = g.exchange_only(10) if n > 1 or args.force_exchange else None
    exch_alt = None
    if (n > 1 or args.force_exchange) and not args.native_rehearsal:
        # the same fragments by the other ways the library can move them
        # (spmv_mgpu.h): the copy engines, and -- ragged fragments -- the
        # three RCCL forms
        exch_alt = {}
        g.set_exchange_engine("copy")
        exch_alt["copy"] = round(g.exchange_only(5), 5)
        g.set_exchange_engine("rccl")
        for kind_x in ("p2p", "bcast", "padded") if ragged else ():
            g.set_ragged_exchange(kind_x)
            exch_alt[kind_x] = round(g.exchange_only(5), 5)
        if ragged:
            g.set_ragged_exchange(args.ragged_exchange)
        else:
            exch_alt["allgather"] = round(g.exchange_only(5), 5)
        g.set_exchange_engine(engine_used)
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
            "rows_per_gpu": Mloc, "logical_shards_per_gpu": L_used,
            "exchange_arrangement": arrangement,
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
            "exchange": ("%d logical shards per device, shard c %s under the "
                         "kernel of c+1" % (
                             L_used, "pushed by the copy engines"
                             if engine_used == "copy" else
                             "all-gathered (staged)"))
            if L_used > 1 else
            ("staged: %d chunks, all-gather of chunk c under the "
             "kernel of c+1" % chunks)
            if chunks > 1 and labels[kernel] not in ("tile_panels", "stream")
            and Mloc % (chunks * 32) == 0 and (n > 1 or args.force_exchange)
            and not ragged
            else "%s (after the kernels; one group)"
            % (args.ragged_exchange if ragged else "allgather"),
            "exchange_ms_alone": round(exch, 5) if exch else None,
            "exchange_alternatives_ms": exch_alt,
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
    g.destroy()
    print(json.dumps(out))



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
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env,
                       timeout=timeout_s)
    if r.returncode != 0:
        raise RuntimeError("native child rc %d: %s" % (r.returncode,
                                                       r.stderr[-400:]))
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
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
            "wall_s": round(time.time() - t0, 1)}

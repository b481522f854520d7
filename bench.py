#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X SpMV engine.

Metric (BASELINE.json): fp64 SpMV GFLOP/s + achieved HBM GB/s (% of the
8 TB/s roofline) at 1/2/4/8 MI355X.  Workload (BASELINE.json configs[2], the
one the >=60 %-of-roofline target is quoted on; configs[4] at 8 GPUs):
synthetic random HLL, hack 32, 32 nnz/row, 10M rows PER GPU (weak scaling),
N = 10M x n_gpus columns, rows partitioned by contiguous ranges, x replicated,
every rank computes its y fragment with the HLL kernel and the fragments are
all-gathered in place over RCCL (torch.distributed backend "nccl").

A step = one SpMV over the whole matrix (+ the all-gather of y when N > 1),
inputs resident in HBM.  value = 2 * nnz_global / step time, in GFLOP/s
(reference definition, include/utils.h:70-75).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ROWS_PER_GPU = 10_000_000
NNZ_PER_ROW = 32
MATRIX_SEED, X_SEED = 42, 7


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows-per-gpu", type=int, default=ROWS_PER_GPU)
    ap.add_argument("--nnz-row", type=int, default=NNZ_PER_ROW)
    ap.add_argument("--window", type=int, default=0,
                    help="column window W of the random family; 0 = N "
                         "(columns anywhere: the worst case)")
    ap.add_argument("--family", default="random",
                    choices=["banded", "random", "ragged", "kkt", "stencil"])
    ap.add_argument("--format", default="hll", choices=["hll", "csr"])
    ap.add_argument("--kernel", type=int, default=-1,
                    help="kernel id (hip_hll.h / hip_csr.h); -1 = default")
    ap.add_argument("--waves", type=int, default=0)
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
    ap.add_argument("--exchange", default="auto", choices=["auto", "halo"],
                    help="halo: only the rows within --halo-rows of another "
                         "rank's range travel (opt-in; NOT the all-gather "
                         "path BASELINE names; for matrices whose columns "
                         "stay near the diagonal)")
    ap.add_argument("--halo-rows", type=int, default=0,
                    help="default: half the column window, rounded up to 32")
    ap.add_argument("--force-exchange", action="store_true",
                    help="initialise RCCL and run the y exchange even with "
                         "one rank (exercises the multi-GPU path on a "
                         "1-GPU box)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    return ap.parse_args()


def cpu_baseline(family, K, W):
    """The reference's own serial + OpenMP CPU path (oracle/_ref/ref_fast,
    built from /root/reference/src by oracle/build_ref.sh) on a bounded
    sample of the same workload; falls back to the oracle port."""
    rows = 2_000_000
    kind = {"banded": 0, "random": 1, "ragged": 2, "kkt": 3,
            "stencil": 4}[family]
    host = os.cpu_count() or 1
    # the GPU box gives one GPU's share of the host (16 cores): time the
    # reference's ladder up to 32 threads and report the best
    ladder = [t for t in (4, 8, 16, 32) if t <= host] or [host]
    cores = max(ladder)
    sample = ("%s %dx%d, %d nnz/row, W=%d (same generator, %d of the %d rows)"
              % (family, rows, rows, K, W, rows, ROWS_PER_GPU))
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_fast")
    env = dict(os.environ, OMP_NUM_THREADS=str(cores), OMP_PROC_BIND="close",
               OMP_PLACES="cores")
    if os.path.exists(ref):
        try:
            out = subprocess.run(
                [ref, "time", str(kind), str(rows), str(rows), str(K), str(W),
                 str(MATRIX_SEED), str(X_SEED), "3"] + [str(t) for t in ladder],
                capture_output=True, text=True, timeout=600, env=env, check=True)
            runs = json.loads(out.stdout)["runs"]
            best = max(runs, key=lambda r: r["gflops"])
            serial = [r for r in runs if r["bench"] == "serial"
                      and r["format"] == "CSR"][0]
            return {"value": round(best["gflops"], 3), "unit": "GFLOP/s",
                    "cores": best["threads"], "kind": "reference",
                    "sample": sample,
                    "best": "%s %s" % (best["format"], best["bench"]),
                    "serial_csr_gflops": round(serial["gflops"], 3),
                    "host_cores": host, "runs": runs}
        except Exception as e:  # pragma: no cover - depends on the box
            err = "ref_fast failed: %r" % (e,)
    else:
        err = "oracle/_ref/ref_fast not present"
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    IRP, JA, AS = O.synth_csr(kind, rows, rows, K, W, MATRIX_SEED)
    x = O.synth_x(X_SEED, 0, rows)
    ms1 = O.time_csr_ms(IRP, JA, AS, x, 1, 3)
    msn = O.time_csr_ms(IRP, JA, AS, x, cores, 3)
    best_ms, thr = (ms1, 1) if ms1 <= msn else (msn, cores)
    return {"value": round(2.0 * len(JA) / (best_ms * 1e6), 3),
            "unit": "GFLOP/s", "cores": thr, "kind": "port", "sample": sample,
            "note": err, "host_cores": host}


# ---- the synthetic workload definition (include/spmv_synth.h) in Python,
# for the in-bench result check: a few rows of y are recomputed from the
# counter-based generator itself (not from the oracle library, which only
# the tests and the cpu_baseline leg may touch).
_M64 = (1 << 64) - 1


def _mix(z):
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def _hash2(seed, a, b):
    return _mix((_mix(seed ^ ((a * 0xD1342543DE82EF95) & _M64)) + b) & _M64)


def _u01(h):
    return (h >> 11) * (1.0 / 9007199254740992.0)


def synth_row_dot(kind, N, K, W, seed, xseed, g):
    """(dot, sum|terms|) of global row g with x, banded (0) / random (1)."""
    if kind == 0:
        st = min(max(g - K // 2, 0), N - K)
        cols = [st + j for j in range(K)]
    elif kind == 1:
        half = W // 2
        lo, hi = max(g - half, 0), min(g + (W - half), N)
        if lo >= hi:
            lo, hi = 0, N
        cols = sorted(lo + int(_u01(_hash2(seed ^ 0x636f6c, g, t)) * (hi - lo))
                      for t in range(K))
    else:
        return None
    acc = sab = 0.0
    for j, c in enumerate(cols):
        v = 2.0 * _u01(_hash2(seed ^ 0x76616c, g, j)) - 1.0
        p = v * _u01(_hash2(xseed, 0x78, c))
        acc += p
        sab += abs(p)
    return acc, sab


def measured_traffic(workload, kname):
    """HBM-side bytes per launch of the dominant kernel from the committed
    rocprofv3 PMC passes (profiles/*.traffic.json, written by
    tools/summarize_profile.py from `tools/profile.sh` runs of THIS command);
    None when no profile of the same workload + kernel is committed."""
    import glob
    best = None
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*.traffic.json"))):
        try:
            t = json.load(open(fn))
        except ValueError:
            continue
        if t.get("workload") == workload and t.get("bench_kernel") == kname:
            best = t
    return best


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import numpy as np
    import torch
    import torch.distributed as dist
    import spmv_scpa_amd as S
    from spmv_scpa_amd import dist as D

    if not torch.cuda.is_available() or S.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X: no GPU visible "
                         "(there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    S.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_exchange
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
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
    kind = {"banded": S.SYNTH_BANDED, "random": S.SYNTH_RANDOM,
            "ragged": S.SYNTH_RAGGED, "kkt": S.SYNTH_KKT,
            "stencil": S.SYNTH_STENCIL}[args.family]
    row0 = rank * Mloc

    # ---- build the shard(s) in HBM (device-side generator + converter) ----
    t_setup = time.time()
    x = torch.empty(Nglob, dtype=torch.float64, device=dev)
    y = torch.zeros(Mglob, dtype=torch.float64, device=dev)
    S.dev_fill_synth(x.data_ptr(), Nglob, X_SEED, 0,
                     torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    tuned = None

    def build_shards(count, rows):
        """`count` logical shards of `rows` rows covering this rank's range"""
        out, nnz, stored = [], 0, 0
        for j in range(count):
            dA = S.CsrDevice.generate(kind, rows, Nglob, K, W, row0 + j * rows,
                                      MATRIX_SEED)
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
    if args.kernel >= 0:
        kernel = args.kernel
    else:
        # kernel chosen by measurement (spmv_*_autotune) on the first shard:
        # the coalesced kernels and, if they run far below the stream rate,
        # the 2-D blocked path.  Every rank must take the same decision.
        kernel, tuned = mat.autotune(x.data_ptr(), y.data_ptr() + 8 * row0)
        if use_dist:
            # rank 0's pick for all: kernel id and, for the blocked path, its
            # schedule and tile height (they decide how the exchange is
            # arranged below: every rank must issue the same collectives)
            sched0 = {"steps": 0, "sweep": 1, "chain": 2}.get(
                mat.panels_schedule(), -1)
            kk = torch.tensor([kernel, sched0, mat.panels_tile_rows() or 0],
                              device=dev)
            dist.broadcast(kk, 0)
            kernel, sched0, tile0 = (int(v) for v in kk.tolist())
            if labels[kernel] == "tile_panels":
                want = ("steps", "sweep", "chain")[sched0]
                if (mat.panels_schedule() != want
                        or (want != "sweep"
                            and mat.panels_tile_rows() != tile0)):
                    mat.build_panels(0, want, tile0)
    nsplit = 2 if args.force_exchange and world == 1 else 4
    if (labels[kernel] == "tile_panels" and use_dist and L == 1
            and Mshard % (nsplit * D.HACK) == 0
            and (mat.panels_schedule() != "sweep" or args.force_exchange)):
        # the blocked path runs whole matrices only: hold the rank's rows as
        # `nsplit` logical shards (4, like the row chunks of the direct
        # kernels) so that the all-gather of one shard runs under the kernel
        # of the next -- at 8 GPUs the exchange (560 MB in per GPU) is longer
        # than the kernel of a matrix with locality.  Not for the sweep
        # schedule: its launch wants every CU (phase counters), RCCL's
        # kernels take some, so the overlap would only delay workgroups --
        # there the exchange follows the kernel.
        model = mat if mat.panels_info() is not None else None
        for m in mats:
            if m is not model:
                m.release()
        L, Mshard = nsplit, Mshard // nsplit
        mats, nnz_local, slots = build_shards(L, Mshard)
        mat = mats[0]
        if model is not None:  # the tuned schedule and tile height
            for m in mats:
                m.build_panels_like(model)
            model.release()
    if labels[kernel] == "tile_panels":
        if mat.panels_info() is None:
            mat.build_panels(0)
        for m in mats[1:]:  # the tuned shard's schedule and tile height
            if m.panels_info() is None:
                m.build_panels_like(mat)
    kname = prefix + labels[kernel]
    pinfo = mat.panels_info() if labels[kernel] == "tile_panels" else None
    launches_per_step = (pinfo["steps"] if pinfo else 1) * L
    # per step and GPU; one launch per logical shard
    alg_bytes = sum(m.algorithmic_bytes for m in mats)
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    chunks = args.chunks if args.chunks > 0 else (4 if world > 1 else 1)
    if labels[kernel] == "tile_panels" or L > 1:
        chunks = 1  # the blocked path runs whole shards only; with logical
        #             shards the shard is the unit of overlap
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
    sharded = D.ShardedSpmv(mats if L > 1 else mat, kernel, rank, world, Mloc,
                            x, y,
                            waves_per_block=args.waves, chunks=chunks,
                            force_exchange=args.force_exchange,
                            mode="halo" if halo else None, halo_rows=halo)

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
    checked = 0
    for g, r in zip(got, rows):
        ref = synth_row_dot(kind, Nglob, K, W, MATRIX_SEED, X_SEED,
                            row0 + int(r))
        if ref is None:
            break
        want, scale = ref
        checked += 1
        if abs(g - want) > 1e-6 * max(abs(want), 1e-3 * scale):
            raise SystemExit("parity check failed on row %d: %r vs %r"
                             % (row0 + r, g, want))

    # ---- warm-up, then EXACTLY K timed steps ----
    for _ in range(args.warmup):
        sharded.step()
    ev = [(torch.cuda.Event(enable_timing=True),
           torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        sharded.step(events=ev[k])
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms = [a.elapsed_time(b) for a, b in ev]

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

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps
    nnz_global = nnz_local * world
    value = 2.0 * nnz_global / (ms_per_step * 1e6)
    kavg = float(np.mean(kern_ms))
    achieved = alg_bytes / (kavg * 1e6)  # GB/s, this rank's kernel

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    wtxt = ("column window W=%s"
            % ("N (anywhere)" if args.window <= 0 else str(W)))
    if args.family == "banded":
        wtxt = "columns s..s+K-1 around the diagonal"
    elif args.family == "stencil":
        wtxt = "grid edge %s" % ("cbrt(N)" if args.window <= 0 else str(W))
    workload = ("%s %s %dx%d per GPU (%dx%d global), hack 32, %d nnz/row, "
                "%s, seed %d"
                % (args.family, args.format.upper(), Mloc, Nglob, Mglob, Nglob,
                   K, wtxt, MATRIX_SEED))
    if L > 1:
        workload += ", %d logical shards of %d rows per GPU" % (L, Mshard)
    traffic = measured_traffic(workload, kname) if world == 1 else None
    out = {
        "metric": "fp64 SpMV GFLOP/s + achieved HBM GB/s (% of roofline), "
                  "1/2/4/8 MI355X",
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
            "workload": workload,
            "kernel": kname,
            "kernel_choice": "autotuned (spmv_%s_autotune)" % args.format
            if tuned is not None else "fixed by --kernel",
            "blocked_schedule": mat.panels_schedule()
            if labels[kernel] == "tile_panels" else None,
            "kernel_launches_per_step": launches_per_step,
            "rows_per_gpu": Mloc, "logical_shards_per_gpu": L,
            "nnz_per_row": K, "nnz_global": nnz_global,
            "stored_slots_per_gpu": slots,
            "partition": "contiguous row ranges, x replicated, in-place "
                         "all-gather(y) over RCCL" if world > 1 else "single GPU",
            "chunks": chunks, "exchange": sharded.mode,
            "exchange_ms_alone": round(exch_ms, 5) if exch_ms else None,
            "halo_rows": halo or None,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": round(traffic["bytes_per_launch"]) if traffic else None,
            "traffic_source": ("profiles/" + traffic["source"]
                               + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                               "separate passes, FETCH_SIZE x2)") if traffic
            else None,
            "kernel": kname,
            "algorithmic_bytes_per_launch": alg_bytes,
            "kernel_ms_avg": round(kavg, 5),
            "kernel_ms_min": round(float(np.min(kern_ms)), 5),
            "kernel_gflops": round(2.0 * nnz_local / (kavg * 1e6), 2),
        },
        "setup_s": round(t_setup, 2),
        "rows_checked": checked,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.family, K, W)
    if world == 1 and L == 1 and not args.no_extras:
        out["extras"] = D.extra_measurements(S, torch, mat, args, x, y, Mloc,
                                             Nglob, K, kind)
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

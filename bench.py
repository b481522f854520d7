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
members of the family side by side: `roofline.variants` carries W = 2^20,
W = 2^17 and the padded [24, 40] rows measured in the same run.

A step = one SpMV over the whole matrix (+ the exchange of y when N > 1),
inputs resident in HBM.  value = 2 * nnz_global / step time, in GFLOP/s
(reference definition, include/utils.h:70-75).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N
    python bench.py --config 4 [--gpus N --partition nnz]   # nlpkkt160-sized .mtx
    python bench.py --config 2     # 1M banded CSR, 16/row, flushed
    python bench.py --native-mgpu --gpus N   # the library's own multi-GPU path

Rank 0 prints the line TWICE: right after the main measurement (the PLAIN
arrangement: each rank's kernel, then one exchange of y) -- complete, flagged
`"provisional": true`, naming `legs_pending` -- and again at the end with what
the optional legs added: `config.exchange_ms_alone` + `exchange_alternatives_ms`,
`config.arrangements` (the overlapped arrangement timed beside the plain one;
`value_best` when it wins), `config.strong` (the fixed 80M x 80M problem),
`config.family_variants` (the banded and W = 2^20 members of the family at this N),
`config.partition_kkt` (even vs nnz-balanced rows on the nlpkkt160-shaped
matrix), `native` (the library's own multi-GPU path, from a fresh child once
the ranks have released the devices); at N = 1 `roofline.variants`,
`cpu_baseline`, `extras`.  The LAST line is the record.  A leg that fails is
named in `legs_failed`; a leg that hangs is ended by a watchdog with the line
printed (benchlib/legs.py); nothing after the main measurement can cost it.

This file is the CLI and the process orchestration; the modes live in
benchlib/: common (workload, result check, roofline records), single
(`--config 2/4`, variants, extras), dist (one rank of the default workload),
native (`--native-mgpu`), cpu (the cpu_baseline leg), launch (rank processes).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.common import *  # noqa: E402,F401,F403
from benchlib.common import (cap_openmp_env, parse_args)  # noqa: E402
from benchlib.cpu import (CPU_WINDOW_MS, cpu_baseline,  # noqa: E402,F401
                          log_cpu_rows, thread_ladder)
from benchlib.launch import (free_port, kfd_gpu_count,  # noqa: E402,F401
                             launch_ranks, rendezvous_only, visible_gpus)


def orchestrate(args, argv):
    """`python bench.py --gpus N` (N > 1) without a launcher.  This parent
    never loads a GPU runtime: it (1) starts the N torch ranks and keeps rank
    0's line, (2) AFTER they have exited starts a fresh child for the
    library's own multi-GPU path and merges its essentials into the line as
    `native`, (3) prints the one merged line.  A failing optional leg goes
    to `legs_failed`; only a failure of the main measurement fails the run."""
    import io
    import time
    want_native = (not args.no_native_leg
                   and args.config == 3 and not args.rendezvous_only
                   and not args.strong and args.shards_per_gpu == 1)
    if want_native:  # the ranks leave `native` to this parent
        os.environ["SPMV_BENCH_PARENT_RUNS_NATIVE"] = "1"
    buf = io.StringIO()
    rc = launch_ranks(args, argv, out=buf)  # rank 0's lines were passed on
    text = buf.getvalue()
    lines = [l for l in text.splitlines() if l.startswith("{")]
    if rc != 0 or not lines or not want_native:
        return rc
    line = json.loads(lines[-1])
    if line.get("provisional"):  # rank 0 never reached its final line
        return rc
    t0 = time.time()
    try:
        from benchlib.native import native_leg
        line["native"] = native_leg(args, args.gpus, timeout_s=180)
    except Exception as e:  # noqa: BLE001 - an optional leg
        line.setdefault("legs_failed", []).append("native_mgpu: %r" % (e,))
        line["native"] = None
    line.setdefault("legs_s", {})["native_mgpu"] = round(time.time() - t0, 1)
    print(json.dumps(line), flush=True)  # the last line is the complete one
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    omp_team = cap_openmp_env()
    # RCCL between processes needs dmabuf IPC on this pool's host driver
    # (exported on the boxes already; set before any runtime loads, for a
    # launcher that starts the ranks with a scrubbed environment)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.native_mgpu:
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
            raise SystemExit("--native-mgpu is ONE process driving --gpus N "
                             "devices: do not start it under a launcher")
        from benchlib.native import native_mgpu_bench
        return native_mgpu_bench(args, argv, omp_team)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(orchestrate(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one rank per "
                         "GPU (python bench.py --gpus N starts them itself)"
                         % (args.gpus, world))
    if args.rendezvous_only:
        raise SystemExit(rendezvous_only(args, rank, world))
    from benchlib.dist import run_rank
    return run_rank(args, argv, omp_team)


# names other tools import from here (tests, tools/): the modes' entry points
def describe_job(*a, **k):
    from benchlib.dist import describe_job as f
    return f(*a, **k)


def extra_measurements(*a, **k):
    from benchlib.single import extra_measurements as f
    return f(*a, **k)


if __name__ == "__main__":
    main()

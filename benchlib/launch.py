"""Process plumbing of `bench.py --gpus N`: counting devices without a GPU
runtime in the parent, starting one rank per GPU, the rendezvous self-test."""
import json
import os
import subprocess
import sys
import time

from .common import METRIC, ROOT


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


def launch_ranks(args, argv, out=None):
    """`python bench.py --gpus N` without a launcher: start N fresh rank
    processes of this script (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, rendezvous on 127.0.0.1), relay rank 0's JSON line, exit
    with the worst return code.  The parent never loads a GPU runtime (devices
    are counted by a child, visible_gpus), so nothing that initialised HIP is
    ever re-executed.  Rank 0's stdout is passed on AS IT ARRIVES (its
    provisional line must survive whatever happens later -- to the ranks or to
    this parent); `out`: a text buffer that receives a copy (bench.py
    orchestrate merges the native leg into the last line)."""
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
            [sys.executable, os.path.join(ROOT, "bench.py")] + list(argv),
            env=e,
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
                sys.stdout.write(chunk.decode(errors="replace"))
                sys.stdout.flush()
        else:
            time.sleep(0.2)
        for r in list(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if r == 0:
                rest = procs[0].stdout.read() or b""
                out0 += rest
                sys.stdout.write(rest.decode(errors="replace"))
                sys.stdout.flush()
            if rc != 0:
                worst = rc if worst == 0 or abs(rc) > abs(worst) else worst
                failed_at = failed_at or time.time()
        # a rank died: the others would wait in a collective forever
        if failed_at and time.time() - failed_at > 20:
            for r in live:
                procs[r].kill()  # exactly the children started above
    if out is not None:
        out.write(out0.decode(errors="replace"))
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


"""The N > 1 control flow of bench.py EXECUTED on a 1-GPU box: two ranks
share the card (`--backend gloo`: RCCL refuses duplicate devices, so the y
fragments are staged through host memory by dist.all_gather_fragments).  The
timings of such a run mean nothing; what it proves is that every line the
driver's 2 / 4 / 8-GPU scaling run will execute has run once: rank 0's pick
broadcast and rebuilt by the other rank, the PLAIN arrangement as the line's
measurement with its provisional line, the overlapped arrangement of each
kernel class as an optional leg beside it (`config.arrangements`,
`value_best`), logical shards, the fixed 80M-style problem of
`config.strong`, the cross-rank result check (each rank verifies rows the
OTHER rank computed), nnz summed over ranks -- and the fault isolation of the
legs: a failure injected on rank 1 only, a rank that hangs in a leg, a job
killed after the main measurement.  It also runs
two persistent sweep launches on one GPU at the same time -- their grids
cannot both be resident, the phase waits must expire and the kernels still
finish (bounded waits: slower, never hung)."""
import json
import os
import subprocess
import sys

import pytest

import spmv_scpa_amd as S

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("extra", [
    # blocked path, sweep schedule: the plain arrangement is the line, the
    # split arrangement an optional leg; config.strong = 4 logical shards
    ["--kernel", "4", "--window", "0"],
    # autotuned pick on a banded matrix (direct kernel): row chunks, staged,
    # as the optional arrangement leg
    ["--kernel", "-1", "--window", "4096"],
    # ragged rows: nnz differs between ranks and is summed, every family's
    # rows are checked through the host generator
    ["--kernel", "-1", "--family", "ragged", "--window", "4096"],
    # the fixed-problem form as the main line: 8 logical shards, 4 per rank
    ["--strong", "--kernel", "4", "--window", "0"],
])
def test_two_ranks_share_one_gpu(extra):
    # no launcher: `python bench.py --gpus 2` starts its own two ranks
    # (bench.py launch_ranks)
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(S.ROOT, "bench.py"),
           "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
           "--rows-per-gpu", "320000", "--no-cpu-baseline", "--no-extras",
           "--kkt-n", "24"]
    r = subprocess.run(cmd + extra, capture_output=True, text=True, env=env,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    # the line is printed right after the main measurement and again at the
    # end; same measurement in both, the last one is the record
    assert len(lines) >= 2 and lines[0]["provisional"] is True
    assert lines[0]["value"] == lines[-1]["value"] > 0
    assert lines[0]["legs_pending"] and "provisional" not in lines[-1]
    j = lines[-1]
    c = j["config"]
    strong = "--strong" in extra
    assert j["n_gpus"] == 2 and j["value"] > 0
    assert j["scaling"] == ("strong" if strong else "weak")
    assert c["backend"].startswith("gloo REHEARSAL")
    assert j["rows_checked"] >= 258 + 3  # own rows + the other rank's
    assert c["rows_per_gpu"] == 320000 * (4 if strong else 1)
    # the line says what joined and on which cards
    rc = c["rccl"]
    assert rc["nranks_joined"] == 2 and len(rc["devices"]) == 2
    assert all(":" in d for d in rc["devices"]), rc  # PCI bus ids
    assert rc["devices"][0] == rc["devices"][1]     # rehearsal: one card
    assert rc["library_links"]                      # RCCL the product links
    roof = j["roofline"]
    assert len(roof["kernel_ms_per_rank"]) == 2
    assert 0 < roof["kernel_ms_min_rank"] <= roof["kernel_ms_max_rank"]
    assert "strong_speedup" in j  # top level; None without a 1-GPU denominator
    # ONE command, ONE complete record: every optional
    # leg present or named as failed / skipped, never a lost line
    assert j["legs_failed"] == [] and j["legs_skipped"] == [], j
    ythr = c["y_rows_per_s"]
    assert ythr["kernel_only"] >= ythr["kernel_then_exchange"] > 0
    assert ythr["measured"] > 0
    kkt = c["partition_kkt"]
    assert kkt["nnz_balanced"]["nnz_max_over_min"] < 1.15 < \
        kkt["even_rows"]["nnz_max_over_min"], kkt
    assert len(kkt["nnz_balanced"]["kernel_ms_per_rank"]) == 2
    assert kkt["nnz_balanced"]["exchange"] == "p2p"
    assert kkt["even_rows"]["exchange"] == "allgather"
    if not strong:
        nat = j["native"]
        assert nat["backend"].startswith("native REHEARSAL"), nat
        assert nat["ms_per_step"] > 0 and len(nat["kernel_ms_per_rank"]) == 2
        assert nat["rows_checked"] >= 2 * 258
        assert "native_mgpu" in j["legs_s"]
        alt = c["exchange_alternatives_ms"]
        assert sorted(alt) == ["bcast", "p2p", "padded"], alt
    if strong:
        # asked for on the command line: 4 logical shards per rank ARE the
        # main arrangement (shard c all-gathered under the kernel of c+1)
        assert c["logical_shards_per_gpu"] == 4 and c["strong"] is None
        assert c["nnz_global"] == 8 * 320000 * 32 and c["exchange"] == "staged"
        assert c["arrangements"] is None and "value_best" not in j
        return
    # the main line is the PLAIN arrangement, whatever the kernel
    assert c["logical_shards_per_gpu"] == 1 and c["chunks"] == 1
    assert c["exchange"] == "allgather"
    assert c["exchange_arrangement"].startswith("plain")
    arr = c["arrangements"]
    assert arr["plain_ms_per_step"] > 0 and arr["alternative_ms_per_step"] > 0
    assert arr["winner"] in ("plain", arr["alternative_kind"])
    assert j["value_best"] >= j["value"] * 0.999
    assert j["ms_per_step_best"] <= j["ms_per_step"] * 1.001
    if arr["winner"] != "plain":
        assert arr["best_ms_per_step"] == j["ms_per_step_best"]
    if "ragged" in extra:
        assert c["nnz_global"] != 2 * 320000 * 32  # summed, not assumed equal
    else:
        assert c["nnz_global"] == 2 * 320000 * 32
    if extra[:2] == ["--kernel", "4"]:
        assert arr["alternative_kind"] == "sweep_split"
        # the banded and the W = 2^20 members of the family at this N
        fam = c["family_variants"]
        assert sorted(fam) == ["W=2^20", "banded"], fam
        for v in fam.values():
            assert v["value_gflops"] > 0 and v["rows_checked"] >= 34
            assert 0 < v["kernel_ms_max_rank"] <= v["ms_per_step"] * 1.05
        assert "banded" in fam["banded"]["workload"]
        st = c["strong"]
        assert "error" not in st, st
        assert st["ms_per_step"] > 0
        # the 1-GPU denominator is a committed full-size measurement: a
        # 320000-row rehearsal has none, and says why
        assert st["speedup_vs_1gpu"] is None and st["one_gpu_source"]
        assert "4 logical shards" in st["problem"]
    else:
        assert arr["alternative_kind"] in ("row_chunks", "logical_shards")
        assert c["family_variants"] is None  # only beside the W = N line


def test_four_ranks_share_one_gpu():
    """the N = 4 line of a scaling run, rehearsed: sweep schedule, the plain
    arrangement measured and the split one timed beside it, config.strong = 2
    logical shards per rank of the fixed problem, each rank checking rows of
    the three others"""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(S.ROOT, "bench.py"),
           "--gpus", "4", "--backend", "gloo", "--steps", "2", "--warmup", "1",
           "--rows-per-gpu", "320000", "--no-cpu-baseline", "--no-extras",
           "--kernel", "4", "--window", "0", "--kkt-n", "24"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    c = j["config"]
    assert j["n_gpus"] == 4 and j["value"] > 0 and j["scaling"] == "weak"
    assert c["rccl"]["nranks_joined"] == 4 and len(c["rccl"]["devices"]) == 4
    assert len(j["roofline"]["kernel_ms_per_rank"]) == 4
    assert j["rows_checked"] >= 258 + 9  # own rows + 3 of each other rank
    assert c["nnz_global"] == 4 * 320000 * 32
    assert c["exchange_arrangement"].startswith("plain")
    assert c["arrangements"]["alternative_kind"] == "sweep_split"
    st = c["strong"]
    assert "error" not in st and "2 logical shards" in st["problem"], st
    assert st["ms_per_step"] > 0


def test_nnz_partition_as_the_main_line_and_a_failing_leg_costs_nothing():
    """`--partition nnz` on a family with ragged rows: ranks own different row
    counts, fragments travel by grouped send / recv, the per-rank entries are
    near-equal; and `bench.py --config 4 --gpus 2 --partition nnz` (the
    nlpkkt160-shaped file over two ranks).  A leg that raises (the native
    child made to fail with an impossible kernel id) is named in legs_failed
    while the rest of the line stands."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    base = [sys.executable, os.path.join(S.ROOT, "bench.py"), "--gpus", "2",
            "--backend", "gloo", "--steps", "2", "--warmup", "1",
            "--no-cpu-baseline", "--no-extras", "--kkt-n", "24"]
    # the hub family: one row of 131 072 entries in rank 0's half moves the cut
    r = subprocess.run(base + ["--rows-per-gpu", "320000", "--family", "hub",
                               "--nnz-row", "6", "--window", "4096",
                               "--format", "csr", "--partition", "nnz",
                               "--kernel", "2"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    c = j["config"]
    assert "nnz-balanced" in c["partition"] and c["exchange"] == "p2p"
    per = c["nnz_per_rank"]
    assert len(per) == 2 and max(per) / min(per) < 1.05, per
    assert c["row_starts"][0] == 0 and c["row_starts"][2] == 640000
    assert c["row_starts"][1] != 320000 and c["row_starts"][1] % 32 == 0
    assert sum(per) == c["nnz_global"] and j["legs_failed"] == []
    assert j["native"]["nnz_per_rank"] == per
    # config 4 over two ranks, nnz-balanced
    r = subprocess.run(base + ["--config", "4", "--partition", "nnz"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["config"]["partition"] == "nnz"
    assert j["config"]["nnz_max_over_min"] < 1.15
    assert j["config"]["exchange"] == "p2p" and j["value"] > 0
    # a failing optional leg: kernel 7 does not exist in the native child's
    # format, the ranks run kernel... the same id fails there too, so break
    # only the child: an unknown ragged exchange cannot be passed on the CLI,
    # a window the native path refuses can -- use --chunks 99 (> 16)
    r = subprocess.run(base + ["--rows-per-gpu", "320000", "--kernel", "1",
                               "--window", "4096", "--chunks", "99",
                               "--no-partition-leg"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["value"] > 0 and j["native"] is None
    assert len(j["legs_failed"]) == 1 and "native_mgpu" in j["legs_failed"][0]


def test_under_torchrun_rank0_starts_the_native_child_itself():
    """The driver launches N > 1 as `python -m torch.distributed.run ...
    bench.py --gpus N`: there is no parent of ours, so rank 0 -- after every
    rank has freed its HBM and the process group is gone -- starts the
    library's own multi-GPU path as a child and merges it into its line."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29581", os.path.join(S.ROOT, "bench.py"),
           "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
           "--rows-per-gpu", "320000", "--window", "65536", "--kkt-n", "24",
           "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["legs_failed"] == [], j["legs_failed"]
    nat = j["native"]
    assert nat and nat["backend"].startswith("native REHEARSAL")
    assert nat["kernel"] == j["config"]["kernel"] or nat["kernel"]
    assert set(j["legs_s"]) >= {"exchange_alone", "exchange_alternatives",
                                "partition_kkt", "native_mgpu"}
    assert j["config"]["partition_kkt"]["speedup_nnz_over_even"] > 0


def _bench(extra, env_extra=None, launcher=False, port="29583"):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE",
                        "SPMV_BENCH_INJECT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.update(env_extra or {})
    head = [sys.executable]
    if launcher:
        head += ["-m", "torch.distributed.run", "--nnodes=1",
                 "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                 "--master-port", port]
    cmd = head + [os.path.join(S.ROOT, "bench.py"),
                  "--gpus", "2", "--backend", "gloo", "--steps", "2",
                  "--warmup", "1", "--rows-per-gpu", "320000",
                  "--no-cpu-baseline", "--no-extras", "--kkt-n", "24"] + extra
    return cmd, env


@pytest.mark.parametrize("launcher", [False, True],
                         ids=["own-ranks", "torchrun"])
def test_a_failure_on_rank_1_only_inside_the_arrangement_leg(launcher):
    """rank 1 alone fails while building the alternative
    arrangement.  Before: rank 0 went on into the leg's collectives and the
    run ended at the process group's 300 s timeout without a line.  Now: the
    ranks agree, both skip the leg, every later leg still runs, rc 0, the
    main line stands and legs_failed names the leg and the rank."""
    cmd, env = _bench(["--kernel", "4", "--window", "0"],
                      {"SPMV_BENCH_INJECT": "arrangement:1:prepare"},
                      launcher)
    r = subprocess.run(cmd, capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines[0]["provisional"] is True
    j = lines[-1]
    assert j["value"] == lines[0]["value"] > 0 and "provisional" not in j
    bad = [f for f in j["legs_failed"] if f.startswith("arrangement")]
    assert len(bad) == 1 and "rank(s) [1]" in bad[0], j["legs_failed"]
    assert len(j["legs_failed"]) == 1
    c = j["config"]
    assert c["arrangements"] is None and "value_best" not in j
    assert c["exchange_ms_alone"] > 0           # the legs before it ...
    assert c["strong"]["ms_per_step"] > 0       # ... and after it ran
    assert c["partition_kkt"]["speedup_nnz_over_even"] > 0
    assert j["native"]["ms_per_step"] > 0


def test_a_rank_that_hangs_inside_a_leg_ends_with_the_line_not_a_timeout():
    """rank 1 never reaches the collectives of the strong leg: the watchdog
    ends the run at the leg's limit with rank 0's final line and rc 0"""
    import time
    cmd, env = _bench(["--kernel", "4", "--window", "0", "--no-native-leg"],
                      {"SPMV_BENCH_INJECT": "exchange_alternatives:1:hang",
                       "SPMV_BENCH_LEG_LIMIT": "12"})
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env,
                       timeout=600)
    assert time.time() - t0 < 200
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    j = lines[-1]
    assert j["value"] == lines[0]["value"] > 0 and "provisional" not in j
    assert any("exchange_alternatives" in f and "limit" in f
               for f in j["legs_failed"]), j["legs_failed"]
    assert j["config"]["exchange_ms_alone"] > 0  # the leg before the hang


def test_a_job_killed_after_the_main_measurement_has_left_its_line():
    """the provisional line is on stdout (flushed) before any optional leg
    starts: kill the whole job while it sits in a leg and parse what is there"""
    import signal
    cmd, env = _bench(["--kernel", "4", "--window", "0"],
                      {"SPMV_BENCH_INJECT":
                       "exchange_alone:0:hang,exchange_alone:1:hang",
                       "SPMV_BENCH_LEG_LIMIT": "600"})
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                         text=True, env=env, start_new_session=True)
    try:
        line = ""
        while not line.startswith("{"):
            line = p.stdout.readline()
            assert line, "bench.py ended without a line"
        j = json.loads(line)
        assert j["provisional"] is True and j["value"] > 0
        assert j["n_gpus"] == 2 and j["roofline"]["frac"] > 0
        assert "exchange_alone" in j["legs_pending"]
    finally:
        os.killpg(p.pid, signal.SIGKILL)  # exactly the session started above
        p.wait()


def test_a_rank_that_never_joins_the_first_exchange_gives_a_failure_record():
    """the MAIN measurement has a deadline too: rank 1 hangs before the first
    exchange of y; the run ends with rc 3 and ONE parsable record that names
    the phase -- not with the process group's 300 s timeout"""
    import time
    cmd, env = _bench(["--kernel", "4", "--window", "0"],
                      {"SPMV_BENCH_INJECT": "main:1:hang",
                       "SPMV_BENCH_MAIN_LIMIT": "20"})
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env,
                       timeout=600)
    assert time.time() - t0 < 200
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = lines[0]
    assert rec["value"] is None and rec["failed"] is True and rec["n_gpus"] == 2
    assert rec["failed_in"].startswith("first step + result check")
    assert "kernel selector" in " ".join(rec["phases_s"])
